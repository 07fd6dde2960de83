"""The tile schedule of k_downdate2 (csrc/ekf_kernels.hip), restated in Python: for every tile count T and CU count the
workgroups' tile lists must cover the lower triangle {(I, J): I >= J} exactly once, a class-A workgroup must end on its
diagonal tile, class B must hold none, and a workgroup never gets more tiles than the host planned.  The HIP code is the
product; this mirrors its index arithmetic (host: rekf_launch_downdate, device: the classA / tri_IJ / cursor logic) so that an
off-by-one there shows up without a GPU.  The GPU suite checks the numbers; this checks the bookkeeping for sizes the GPU
tests do not reach (T up to 64 = n up to 4096)."""
import math

import pytest


def host_plan(T, slots=256):
    """downdate_schedule (csrc/ekf_kernels.hip): (grid, lo, x, sub) -- class-B workgroups take lo tiles, the first x one more."""
    room = slots - T if slots - T > 1 else 1
    nB = (T - 1) * (T - 2) // 2
    sub = 2
    if (nB + room - 1) // room < 3:
        sub = 1
        nB = T * (T - 1) // 2
    lo, x = nB // room, nB % room
    grid = T + (room if lo > 0 else x)
    if grid >= 64:
        grid = (grid + 7) & ~7
    return grid, lo, x, sub


def kernel_plan(T, grid):
    """What k_downdate2 derives itself when the host only has a bound of n (dd_sub = 0)."""
    room = grid - T if grid - T > 1 else 1
    nB = (T - 1) * (T - 2) // 2
    sub = 2
    if (nB + room - 1) // room < 3:
        sub = 1
        nB = T * (T - 1) // 2
    return nB // room, nB % room, sub


def device_tiles(T, grid, lo, x, sub, block):
    """Tiles of workgroup `block`, in processing order."""
    w = block
    if grid >= 8 and grid % 8 == 0:
        w = (block & 7) * (grid >> 3) + (block >> 3)          # an XCD's workgroups take consecutive ranges
    if w < T:                                                  # class A
        tiles = []
        if sub == 2 and w + 1 < T:
            tiles.append((w + 1, w))
        tiles.append((w, w))
        return tiles
    TT = T - sub
    nB = (T - sub + 1) * (T - sub) // 2
    wq = w - T
    t0 = wq * lo + min(wq, x)
    t1 = t0 + lo + (1 if wq < x else 0)
    t0, t1 = min(t0, nB), min(t1, nB)
    if t0 >= t1:
        return []
    # cursor seeded by the closed form, then walked (tri_IJ)
    bq = 2.0 * TT + 1.0
    Jg = int((bq - math.sqrt(max(bq * bq - 8.0 * t0, 0.0))) * 0.5)
    Jg = max(0, min(TT - 1, Jg))
    cur_J, cur_c0 = Jg, Jg * TT - (Jg * (Jg - 1)) // 2
    out = []
    for tt in range(t0, t1):
        while cur_J + 1 < TT and tt >= cur_c0 + (TT - cur_J):
            cur_c0 += TT - cur_J
            cur_J += 1
        while cur_J > 0 and tt < cur_c0:
            cur_J -= 1
            cur_c0 -= TT - cur_J
        out.append((cur_J + (tt - cur_c0) + sub, cur_J))
    return out


@pytest.mark.parametrize("slots", [256, 304, 64, 239, 223, 199])     # (the odd ones: the downdate as a ROLE of k_mid's grid, 256 - K - mid workgroups)
def test_every_lower_triangle_tile_exactly_once(slots):
    for T in range(1, 65):
        grid, lo, x, sub = host_plan(T, slots)
        seen = {}
        busy = 0
        for b in range(grid):
            tl = device_tiles(T, grid, lo, x, sub, b)
            busy += bool(tl)
            assert len(tl) <= max(lo + 1, 2), (T, b, tl)
            diag = [t for t in tl if t[0] == t[1]]
            assert len(diag) <= 1 and (not diag or tl[-1] == diag[0]), (T, b, tl)      # the diagonal tile comes last
            for t in tl:
                assert t not in seen, (T, t, b, seen[t])
                seen[t] = b
        want = {(i, j) for i in range(T) for j in range(i + 1)}
        assert set(seen) == want, (T, sorted(want - set(seen))[:5], sorted(set(seen) - want)[:5])
        if T >= 24 and slots >= T + 8:
            assert busy >= min(slots, len(want) // 2) - 8, (T, busy)                   # no CU is left idle once there is work for all


def test_host_bound_of_T_may_exceed_the_kernels():
    """The host sizes the grid from an upper bound of n (T_host >= T_kernel) and then passes no schedule: the kernel derives
    (lo, x, sub) from its own T and the grid it finds; its own T must still be covered exactly once."""
    for T_k in range(1, 40):
        for T_h, slots in [(T_k, 256), (T_k + 1, 256), (T_k + 7, 256), (2 * T_k, 256), (T_k + 1, 223), (T_k + 3, 199)]:
            grid, _, _, _ = host_plan(T_h, slots)
            lo, x, sub = kernel_plan(T_k, grid)
            seen = set()
            for b in range(grid):
                for t in device_tiles(T_k, grid, lo, x, sub, b):
                    assert t not in seen
                    seen.add(t)
            assert seen == {(i, j) for i in range(T_k) for j in range(i + 1)}, (T_k, T_h)


def queue_items(T, run=3):
    """The work items of the downdate ROLE inside k_mid's grid (dd_body<KC, true>, round 5): items 0 .. T-1 are class A (the tile below
    diagonal tile w, then that tile), then runs of `run` tiles of the triangle I >= J + 2, column by column; whoever is free takes the
    next item from RekfCtl::dd_queue."""
    nB = (T - 1) * (T - 2) // 2
    n_items = T + (nB + run - 1) // run
    items = []
    for it in range(n_items):
        if it < T:
            items.append(device_tiles(T, T + 1, 0, 0, 2, it) if False else ([(it + 1, it)] if it + 1 < T else []) + [(it, it)])
        else:
            t0, t1 = run * (it - T), min(run * (it - T) + run, nB)
            TT = T - 2
            out = []
            for tt in range(t0, t1):
                J = 0
                c0 = 0
                while tt >= c0 + (TT - J):
                    c0 += TT - J
                    J += 1
                out.append((J + (tt - c0) + 2, J))
            items.append(out)
    return items


def test_the_queue_hands_every_tile_out_exactly_once():
    for T in range(1, 70):
        seen = set()
        for tl in queue_items(T):
            assert 1 <= len(tl) <= 3
            diag = [t for t in tl if t[0] == t[1]]
            assert len(diag) <= 1 and (not diag or tl[-1] == diag[0])      # a diagonal tile ends its item (the mirror reuses the panels' LDS)
            for t in tl:
                assert t not in seen, (T, t)
                seen.add(t)
        assert seen == {(i, j) for i in range(T) for j in range(i + 1)}, T
