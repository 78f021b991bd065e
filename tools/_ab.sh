python -m pytest tests -x -q -m gpu -k "reid or conv or small_graph or pipeline_inject" 2>&1 | tail -2
cp ai-camera_amd/libaicam.so /tmp/new.so
for i in 1 2; do
for v in new old; do
  if [ $v = new ]; then cp /tmp/new.so ai-camera_amd/libaicam.so; else cp tools/_old_libaicam.so ai-camera_amd/libaicam.so; fi
  for shape in "32 16 128 128" "16 8 256 256" "8 4 512 512"; do
    echo -n "$v shape $shape res=0: "
    CB_NET=1 python tools/conv_bench.py $shape 3 15360 8 0 2>&1 | tail -1 | sed -e 's/env=.*//' | cut -c125-200
  done
done
done
cp /tmp/new.so ai-camera_amd/libaicam.so
