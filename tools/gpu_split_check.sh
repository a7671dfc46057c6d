# the sharded subset in the serial form, with the split form forced on small swarms, and config 5 at size (split by itself)
OUT=gpurun_out/split; rm -rf $OUT; mkdir -p $OUT
echo "== serial form (MRS_SHARD_SPLIT=0)"; MRS_SHARD_SPLIT=0 timeout -k 10 400 python -m pytest tests/test_export_sets_gpu.py -x -q -m gpu > $OUT/serial.log 2>&1 || { tail -40 $OUT/serial.log | cut -c1-400; exit 1; }
tail -2 $OUT/serial.log
echo "== split form forced (MRS_SHARD_SPLIT_MIN_BLOCKS=1)"; MRS_SHARD_SPLIT_MIN_BLOCKS=1 timeout -k 10 400 python -m pytest tests/test_export_sets_gpu.py -x -q -m gpu -s > $OUT/split.log 2>&1 || { tail -60 $OUT/split.log | cut -c1-400; exit 1; }
tail -2 $OUT/split.log
echo "== config 5 at size"; timeout -k 10 500 python -m pytest tests/test_config5_gpu.py -x -q -m gpu -s > $OUT/config5.log 2>&1 || { tail -60 $OUT/config5.log | cut -c1-400; exit 1; }
grep -E "config 5:|passed|failed" $OUT/config5.log | cut -c1-700
