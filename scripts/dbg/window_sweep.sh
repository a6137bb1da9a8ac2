mkdir -p gpurun_out/r4f
for D in 768 384; do for W in 1 2 4 16 100000; do
  echo "dim $D window $W: $(DIM=$D Q=1024 YAMS_ACCEL_I8R_WINDOW=$W timeout 200 python scripts/dbg/filter_forms.py 80 2>/dev/null | tail -1)" | tee -a gpurun_out/r4f/window.txt
done; done
timeout 200 python scripts/dbg/upload_rates.py > gpurun_out/r4f/upload.json 2> gpurun_out/r4f/upload.err; cat gpurun_out/r4f/upload.json
