mkdir -p gpurun_out/r04o
for kv in "lanes=4" "lanes=3" "lanes=5" "lanes=6" "patch_wgs=96" "patch_wgs=160" "patch_wgs=256" "patch_gemm_wgs=96" "patch_gemm_wgs=160" "patch_gemm_wgs=256" "wide_gemm=0" "wide_gemm=1" "patch_gemm=3" "lanes=4"; do
  if [[ $kv == lanes=* ]]; then extra="--lanes ${kv#lanes=}"; knob=""; else extra=""; knob="--knob $kv"; fi
  python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-detect --no-extra-legs $extra $knob > gpurun_out/r04o/b_$kv.json 2>> gpurun_out/r04o/bench.err
  python -c "import json;d=json.load(open('gpurun_out/r04o/b_$kv.json'));print('$kv', d['value'], d['value_cold'])"
done
