#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (gpurun_out/pmc/*_counter_collection.csv) per libscs kernel.
Only full-size launches (largest grid of each kernel) are averaged.  HBM traffic follows MI355X_MICROARCH.md
(HBM section): FETCH_SIZE is in KiB and reads exactly half of a wide coalesced stream on gfx950 -> x2;
WRITE_SIZE in KiB as is."""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    m = re.search(r'ConvCfg<([^>]*)>(, true)?', name)
    if m:               # (", true": the guarded instantiation behind the split kernels, scs_api.hip forward_chunk_split - its own row)
        return ('guarded ' if m.group(2) else '') + 'conv<' + m.group(1).replace(' ', '') + '>'
    return name.split('(')[0].replace('scs::', '')


def main(pattern, out_json=None):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for fn in sorted(glob.glob(pattern)):
        for r in csv.DictReader(open(fn)):
            if 'scs::' not in r['Kernel_Name']:
                continue
            k = short(r['Kernel_Name'])
            agg[k][r['Counter_Name']].append((float(r['Counter_Value']), int(r['Grid_Size'])))
            dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp']), int(r['Grid_Size'])))
    res = {}
    for k, v in agg.items():
        row = {}
        for c, vals in v.items():
            mx = max(g for _, g in vals)
            full = [x for x, g in vals if g == mx]
            row[c] = sum(full) / len(full)
        mx = max(g for _, g in dur[k])
        d = [x for x, g in dur[k] if g == mx]
        row['grid'] = mx
        row['dur_us_under_pmc'] = sum(d) / len(d) / 1e3
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in row and 'SQ_BUSY_CYCLES' in row and row['SQ_BUSY_CYCLES']:
            # BUSY_CYCLES is summed over the SEs/XCDs; MFMA busy over SIMDs: report the ratio to wave-cycles too
            row['mfma_busy_per_wave_cycle'] = row['SQ_VALU_MFMA_BUSY_CYCLES'] / (4.0 * row['SQ_WAVE_CYCLES']) if row.get('SQ_WAVE_CYCLES') else None
        if 'SQ_LDS_IDX_ACTIVE' in row and row['SQ_LDS_IDX_ACTIVE']:
            row['lds_conflict_frac'] = row['SQ_LDS_BANK_CONFLICT'] / row['SQ_LDS_IDX_ACTIVE']
        if 'FETCH_SIZE' in row:
            row['hbm_read_bytes_corrected'] = row['FETCH_SIZE'] * 1024 * 2
        if 'WRITE_SIZE' in row:
            row['hbm_write_bytes'] = row['WRITE_SIZE'] * 1024
        res[k] = row
    for k, row in res.items():
        print(k)
        for c in sorted(row):
            print('   %-30s %.6g' % (c, row[c]) if isinstance(row[c], (int, float)) else '   %-30s %s' % (c, row[c]))
    if out_json:
        json.dump(res, open(out_json, 'w'), indent=1)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
