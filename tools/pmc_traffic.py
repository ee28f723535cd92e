#!/usr/bin/env python3
"""profiles/traffic.json from rocprofv3 --pmc passes: HBM bytes per launch of the conv kernels bench.py may name as dominant.

  python tools/pmc_traffic.py <forward pmc summary json> <train pmc summary json> <csrc sha16 of the measured tree> [commit]

The two inputs are what tools/pmc_summary.py writes from the FETCH_SIZE and WRITE_SIZE passes (separate passes, FETCH_SIZE
x 2 on gfx950, KiB -> bytes: MI355X_MICROARCH.md, HBM section) of `bench.py --no-train-block` and `bench.py --mode train`.
The sha16 is bench.csrc_sha16() printed ON THE BOX THAT MEASURED (gpurun_out/csrc_sha16.txt); bench.py reports a traffic
figure only while the kernel sources still hash to it.
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# stage name of bench.py -> regular expression on the kernel's short name (tools/pmc_summary.py: short())
FORWARD = {
    'conv1+conv2+pool1': r'^conv<20,20,30,1,1,1,',
    'conv3': r'^conv<20,40,14,\d+,0,0,',
    'conv4+pool2': r'^conv<40,40,12,\d+,1,0,',
    'conv5': r'^conv<40,60,5,\d+,0,0,',
}
# the split kernels of the same stages (option fwd_split: csrc/conv_split.h, dense_split.h); keys carry the prefix 'split: '
SPLIT = {
    'conv1+conv2+pool1': r'conv12_split_kernel<true, false>',
    'conv3': r'conv_split_all_kernel<SpCfg<20, 40, 14,',
    'conv4+pool2': r'conv_split_all_kernel<SpCfg<40, 40, 12,',
    'conv5': r'conv_split_kernel<SpCfg<40, 60, 5,',
    'd1x3 / FC1 (one kernel, two launches: the average)': r'dense_split_kernel<DsCfg<17,',
    'fc_2': r'dense_split_kernel<DsCfg<18,',
}
TRAIN = {
    'conv1+conv2 fwd': r'^conv<20,20,30,1,3,8,',
    'conv1+conv2 fwd (train_split)': r'conv12_split_train_kernel',
    'conv3 fwd': r'^conv<20,40,14,1,3,[45],',
    'conv4 fwd': r'^conv<40,40,12,1,3,4,',
    'conv5 fwd': r'^conv<40,60,5,\d+,3,5,',
    'conv5 dgrad': r'^conv<60,40,7,',
    'conv5 backward, one launch (dgrad + wgrad + bn4 bwd sums)': r'conv5_backward_kernel',
    'conv4 dgrad': r'^conv<40,40,14,',
    'conv3 dgrad': r'^conv<40,20,16,',
    'conv2 dgrad + layer-1 sums': r'^conv<20,20,32,',
    'conv5 wgrad': r'wgrad_mfma_kernel<WgCfg<40, 60, 5,',
    'conv4 wgrad': r'wgrad_mfma_kernel<WgCfg<40, 40, 12,',
    'conv3 wgrad': r'wgrad_mfma_kernel<WgCfg<20, 40, 14,',
    'conv2 wgrad': r'wgrad_mfma_kernel<WgCfg<20, 20, 30,',
}


def pick(summary, pattern):
    for k, row in summary.items():
        if re.search(pattern, k) and 'hbm_read_bytes_corrected' in row and 'hbm_write_bytes' in row:
            return row['hbm_read_bytes_corrected'] + row['hbm_write_bytes'], row['hbm_read_bytes_corrected'], row['hbm_write_bytes']
    return None


def main():
    fwd = json.load(open(sys.argv[1]))
    trn = json.load(open(sys.argv[2]))
    sha = sys.argv[3].strip()
    commit = sys.argv[4] if len(sys.argv) > 4 else subprocess.check_output(['git', '-C', ROOT, 'rev-parse', '--short', 'HEAD']).decode().strip()
    out = {'_csrc_sha16': sha, '_source_commit': commit,
           '_note': 'HBM bytes per launch (bench.py defaults: 25 000 patches per forward launch, batch 256 per training launch): rocprofv3 '
                    '--pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md '
                    'HBM section), KiB -> bytes; tools/pmc_summary.py + tools/pmc_traffic.py', '_detail': {}}
    for name, pat in FORWARD.items():
        r = pick(fwd, pat)
        if r:
            out[name] = r[0]
            out['_detail'][name] = {'read': r[1], 'write': r[2]}
    for name, pat in SPLIT.items():
        r = pick(fwd, pat)
        if r:
            out['split: ' + name] = r[0]
            out['_detail']['split: ' + name] = {'read': r[1], 'write': r[2]}
    for name, pat in TRAIN.items():
        r = pick(trn, pat)
        if r:
            out['train: ' + name] = r[0]
            out['_detail']['train: ' + name] = {'read': r[1], 'write': r[2]}
    json.dump(out, open(os.path.join(ROOT, 'profiles', 'traffic.json'), 'w'), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
