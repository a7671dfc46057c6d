#!/bin/bash
# packed state download written by the kernel straight into pinned memory for few UAVs: facade tests + latency A/B
python -m pytest tests/test_facade_cpp.py tests/test_pool_primitives_gpu.py tests/test_l0_classes.py -x -q -m gpu 2>&1 | tail -3
for v in 1024 0; do echo "MRS_DIRECT_STATE_MAX=$v"; MRS_DIRECT_STATE_MAX=$v ./tests/cpp/facade_loop_test | grep "LATENCY\|TICK_US"; MRS_DIRECT_STATE_MAX=$v ./tests/cpp/facade_loop_test single | grep "LATENCY"; done
for v in 1024 0; do echo "MRS_DIRECT_STATE_MAX=$v"; MRS_DIRECT_STATE_MAX=$v ./tests/cpp/facade_loop_test | grep "LATENCY\|TICK_US"; MRS_DIRECT_STATE_MAX=$v ./tests/cpp/facade_loop_test single | grep "LATENCY"; done
