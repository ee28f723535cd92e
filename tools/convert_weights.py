#!/usr/bin/env python3
"""
One-time converter (runs in the authoring container only, where /root/reference is mounted):

  nets/miccai2012_v1/miccai2012_v1.pkl          -> sub-cortical_segmentation_amd/nets/miccai2012_v1/miccai2012_v1.{f32,json}
  nets/miccai2012_v1/miccai2012_v1_history.pkl  -> tests/golden/miccai2012_v1_history.json

Both pickles are pure data (OrderedDict / list of numpy arrays & scalars); they are read with the
allow-listed unpickler in cnn_cort/params.py.  The outputs are data files (raw Lasagne layout,
nothing folded or flipped), so training can round-trip through them.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'sub-cortical_segmentation_amd'))
from cnn_cort import params as P  # noqa: E402

REF = '/root/reference/nets/miccai2012_v1'


def main():
    w = P.load_params(os.path.join(REF, 'miccai2012_v1.pkl'))
    out = os.path.join(ROOT, 'sub-cortical_segmentation_amd', 'nets', 'miccai2012_v1', 'miccai2012_v1')
    os.makedirs(os.path.dirname(out), exist_ok=True)
    print('wrote', P.save_params(out, w), P.N_STORED, 'floats')

    hist = P.read_reference_pickle(os.path.join(REF, 'miccai2012_v1_history.pkl'))
    rows = [{k: (bool(v) if k.endswith('_best') else (int(v) if k == 'epoch' else float(v)))
             for k, v in sorted(r.items())} for r in hist]
    dst = os.path.join(ROOT, 'tests', 'golden', 'miccai2012_v1_history.json')
    with open(dst, 'w') as f:
        json.dump(rows, f, indent=0)
    print('wrote', dst, len(rows), 'epochs')


if __name__ == '__main__':
    main()
