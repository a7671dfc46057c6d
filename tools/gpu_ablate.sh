set -x
mkdir -p gpurun_out/r2g
for b in 0 1 2 4 8 16 31; do
  timeout -k 10 300 bash tools/variant_bench.sh "-DMRS_ABL=$b" --workload position+collisions --steps 1000 >> gpurun_out/r2g/ablate.log 2>> gpurun_out/r2g/ablate.err
done
cat gpurun_out/r2g/ablate.log
