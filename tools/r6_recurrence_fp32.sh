# Round 6 (GPU box): the recurrence tests of tests/test_gpu_fullsize.py on the exact-fp32 mode's Winograd kernels (gate 1e-4)
out=gpurun_out/r06_recurrence_fp32_winograd.txt; : > $out
for t in test_drift_100_frames_346x260_8_sequences test_one_sequence_split_k_100_frames_346x260 test_recurrence_at_the_64_sequence_dispatch test_e2vid_640x480_vs_oracle; do
  EVR_FP32=1 timeout 1500 python -m pytest tests/test_gpu_fullsize.py::$t -q -s -x -m gpu -p no:cacheprovider 2>&1 | grep -E "worst per-pixel|passed|failed|Error" | sed "s/^/EVR_FP32=1 $t: /" | tee -a $out
done
EVR_FP32=1 EVR_WINO=0 timeout 1500 python -m pytest tests/test_gpu_fullsize.py::test_drift_100_frames_346x260_8_sequences -q -s -x -m gpu -p no:cacheprovider 2>&1 | grep -E "worst per-pixel|passed|failed" | sed "s/^/EVR_FP32=1 EVR_WINO=0 (direct form) drift_100x8: /" | tee -a $out
