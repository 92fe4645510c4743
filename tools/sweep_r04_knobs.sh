# knob sweep of round 4 (run through gpurun from the repo root): bench.py --steps 20 --warmup 5, one line per setting
mkdir -p gpurun_out/r04p
for kv in "lanes=4" "patch_wgs=64" "patch_wgs=96" "patch_wgs=112" "patch_gemm_wgs=64" "patch_gemm_wgs=96" "patch_gemm_wgs=112" "lanes=4" "stft_pair_rows2=0"; do
  if [[ $kv == lanes=* ]]; then extra="--lanes ${kv#lanes=}"; knob=""; else extra=""; knob="--knob $kv"; fi
  python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-detect --no-extra-legs $extra $knob > gpurun_out/r04p/b_$kv.json 2>> gpurun_out/r04p/bench.err
  python -c "import json;d=json.load(open('gpurun_out/r04p/b_$kv.json'));print('$kv', d['value'], d['value_cold'], [(x['kernel'], round(x['ms_per_launch']*1e3,1)) for x in d['kernels']][:3])"
done
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-detect --no-extra-legs --knob patch_wgs=96 --knob patch_gemm_wgs=96 > gpurun_out/r04p/b_both96.json 2>> gpurun_out/r04p/bench.err
python -c "import json;d=json.load(open('gpurun_out/r04p/b_both96.json'));print('both96', d['value'], d['value_cold'])"
