#!/bin/bash
# frame time of bench.py --workload caustic_sppm for the in-tree library and every _diag/lib_*.so (compile-time tuning A/B)
run() { python bench.py --workload caustic_sppm --no-traffic --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['roofline']['kernel_ms_per_step'])"; }
echo "default: $(run)"
for L in _diag/lib_*.so; do echo "$(basename $L): $(TRHIP_LIB=$PWD/$L run)"; done
