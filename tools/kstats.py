"""Per-step view of a rocprofv3 --kernel-trace --stats CSV: python tools/kstats.py <kernel_stats.csv> <steps+warmup>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = 0.0
for r in rows:
    n = r['Name']
    if 'at::' in n or '__amd_rocclr' in n and False:
        continue
    us = float(r['TotalDurationNs']) / steps / 1000
    tot += us
    print('%8.1f us/step %5.1f calls/step  avg %7.1f us  %s' % (us, int(r['Calls']) / steps, float(r['AverageNs']) / 1000, n[:120]))
print('total %.1f us/step' % tot)
