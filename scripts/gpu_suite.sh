#!/bin/bash
# the whole GPU suite, as the driver runs it (output kept under gpurun_out/)
mkdir -p gpurun_out
python -m pytest tests/ -x -q -m gpu > gpurun_out/suite.txt 2>&1
tail -12 gpurun_out/suite.txt
