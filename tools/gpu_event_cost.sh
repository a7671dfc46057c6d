# cross-stream dependency cost for the interior/boundary split (tools/event_pingpong.hip), three duty points
OUT=gpurun_out/evcost; rm -rf $OUT; mkdir -p $OUT
hipcc -O2 --offload-arch=gfx950 tools/event_pingpong.hip -o $OUT/event_pingpong || exit 1
timeout -k 10 120 $OUT/event_pingpong 22 6 20 > $OUT/log.txt 2>&1 || { cat $OUT/log.txt; exit 1; }
timeout -k 10 120 $OUT/event_pingpong 22 6 10 >> $OUT/log.txt 2>&1 || { cat $OUT/log.txt; exit 1; }
timeout -k 10 120 $OUT/event_pingpong 22 6 30 >> $OUT/log.txt 2>&1 || { cat $OUT/log.txt; exit 1; }
grep -v "ReleaseToDevice" $OUT/log.txt
