echo "== long soak on the final build of the round (persistent histogram kind, direct statistics, early uploads, sums in two groups)"
python tools/gpu_soak.py 23 150 2>&1 | tail -1
python tools/batch_fuzz.py 29 200 2>&1 | tail -1
SJPEG_HIP_BATCH_PARTS=3 python tools/batch_fuzz.py 31 60 2>&1 | tail -1
SJPEG_HIP_HISTO_SLOTS=5 python tools/batch_fuzz.py 37 60 2>&1 | tail -1
python tools/low_entropy_fuzz.py 41 60 2>&1 | tail -1
python tools/extremes_fuzz.py 43 60 2>&1 | tail -1
python tools/many_frames_check.py 2>&1 | tail -5
