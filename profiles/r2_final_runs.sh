python profiles/spectral_rate.py 1024 64 > gpurun_out/r2f_spectral.log 2>&1
python bench.py --workload dr_interior --no-cpu-baseline --no-also --no-build > gpurun_out/r2f_dri.log 2>&1
bash profiles/share.sh > gpurun_out/r2f_share.log 2>&1
python profiles/build_time.py > gpurun_out/r2f_build.log 2>&1
for l in 1 2 3; do echo -n "cornell layout $l: "; python bench.py --steps 3 --warmup 1 --spp 256 --accel-layout $l --no-cpu-baseline --no-also --no-build 2>&1 | grep -o "\"value\": [0-9.]*" | head -1; done > gpurun_out/r2f_layout.log 2>&1
echo done
