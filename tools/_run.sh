#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/r4b_pytest_gpu_full_suite.txt 2>&1 < /dev/null; tail -3 $O/r4b_pytest_gpu_full_suite.txt | head -1
timeout 1500 python tools/soak_hybrid.py --scenes 80 --rays 400000 --frames 48 --seed 5 > $O/r4b_soak_hybrid.txt 2>$O/r4b_soak_hybrid.err < /dev/null; tail -1 $O/r4b_soak_hybrid.txt
timeout 600 python tools/soak_film.py --cases 100 > $O/r4b_soak_film.txt 2>&1 < /dev/null; tail -1 $O/r4b_soak_film.txt
timeout 900 python tools/soak_path.py --scenes 60 > $O/r4b_soak_path.txt 2>&1 < /dev/null; tail -1 $O/r4b_soak_path.txt
timeout 900 python tools/soak_sppm.py --scenes 40 > $O/r4b_soak_sppm.txt 2>&1 < /dev/null; tail -1 $O/r4b_soak_sppm.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/final_runs.sh r4b < /dev/null
