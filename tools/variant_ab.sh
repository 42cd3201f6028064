#!/bin/bash
# Development tool: interleaved A/B of two tuning variants through bench.py (usage: variant_ab.sh A B [rounds])
A=${1:-0}; B=${2:-7}; R=${3:-3}
for r in $(seq $R); do for v in $A $B; do python bench.py --no-cpu-baseline --variant $v --steps 300 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('variant', $v, 'step us', round(d['ms_per_step']*1e3,2), 'kernel us (events)', round(d['roofline']['kernel_ms_mean']*1e3,2), '(device clock)', round(d['roofline']['kernel_ms_mean_device_clock']*1e3,2), 'rows/s', '%.4g' % d['value'])"; done; done
