#!/usr/bin/env python
"""Replay a rekf-dump-v1 file (reflector_ekf_slam_amd/node_replay.py documents the schema) through the HIP path the way the
reference's Node drives its components: 2D detector -> EKF (with options.map_path) -> MapBuilder::AddRangeData, then
SaveReflectorResult.  On an MI355X:

    python scripts/replay_dump.py --synth /tmp/demo.npz          # write a synthetic dump (no bag blob ships with the reference)
    python scripts/replay_dump.py /tmp/demo.npz --save /tmp/reflector_map
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dump", nargs="?")
    ap.add_argument("--synth", metavar="PATH", help="write a synthetic dump here and exit")
    ap.add_argument("--save", metavar="FILEBASE", help="SaveReflectorResult(filebase): writes filebase.txt")
    ap.add_argument("--max-landmarks", type=int, default=64, help="initial capacity (grows on demand)")
    a = ap.parse_args()
    from reflector_ekf_slam_amd import node_replay as NR
    from reflector_ekf_slam_amd import synth
    if a.synth:
        cfg = synth.SessionConfig("dump_demo", 40, 12, synth.DIFF, seed=31, speed=1.0, row_spacing=6.0)
        d = NR.synth_dump(a.synth, cfg, max_scans=80)
        print(f"wrote {a.synth}: {d.odom_t.shape[0]} odometry messages, {d.scan_t.shape[0]} scans, {d.ranges.shape[0]} beams")
        return
    d = NR.read_dump(a.dump)
    t0 = time.time()
    node = NR.replay(d, NR.hip_backend(max_landmarks=a.max_landmarks))
    dt = time.time() - t0
    n = node.slam.n if node.slam is not None else 0
    matched = sum(p is not None for p in node.log.match_poses)
    print(f"{d.scan_t.shape[0]} scans, {d.odom_t.shape[0]} odometry messages in {dt:.2f} s; state dimension {n} "
          f"({(n - 3) // 2} reflectors); {matched} scans matched into the grid; last pose {node.log.path[-1][1:] if node.log.path else None}")
    if a.save:
        print("saved", node.SaveReflectorResult(a.save))


if __name__ == "__main__":
    main()
