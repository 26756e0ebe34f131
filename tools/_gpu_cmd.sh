out=gpurun_out/big3; mkdir -p $out
for rep in 1 2; do for lib in build/ab/libsar_hip_base.so strange_attractor_renderer_amd/libsar_hip.so; do
  echo "== $lib"; SAR_LIBRARY=$PWD/$lib timeout 200 python tools/config_table.py --only C2 C3 C4/8 X4K X2560 XHD --reps 5 --out $out/t.jsonl 2>&1 | grep -o '"config": "[^"]*".*"fold_ms": [0-9.]*' | sed 's/"jobs.*wall_ms/ wall_ms/'
done; done
