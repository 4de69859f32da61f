#!/bin/bash
# Runs ON THE GPU BOX: is the slow process mode (6.0-6.3 instead of 5.7 us per step) in the kernels or between them?
# N fresh processes of tools/span_gap.py on the light span build (production kernel + start / acknowledged stamps);
# prints per process the HIP-event cadence, the kernel span, the inter-kernel gap and the wave-life percentiles.
N=${1:-12}
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq 1 $N); do
  MSNAKE_LIB=$R/self-play-on-multi-snakes-environment_amd/libmsnake_span.so timeout -k 10 120 python $R/tools/span_gap.py 4096 512 2>/dev/null | python3 -c "
import sys, json
d = json.load(sys.stdin); s = d['summary']; r = d['regions'][2]
print(json.dumps({'hip_event_us': s['hip_event_us_per_launch'], 'span_us': s['span_us'], 'gap_us': s['gap_us'], 'wave_life_p50': r['wave_life_us']['p50'],
                  'wave_life_p99': r['wave_life_us']['p99'], 'first_to_last_start': r['first_to_last_wave_start_us'], 'last_ack_per_xcd': r['last_ack_per_xcd_us_since_launch_start']}))"
done
