"""PCIe-inclusive rate of the literal reference API: net.predict_proba on HOST arrays (1.23 GB uploaded per 100 000
patches).  python tools/host_rate.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'sub-cortical_segmentation_amd'))
sys.path.insert(0, ROOT)
from cnn_cort import synth
from cnn_cort.nets import build_model

opts = {'experiment': 'miccai2012_v1', 'net_verbose': 0, 'train_split': 0.25, 'max_epochs': 1, 'patience': 20,
        'patch_size': [32, 32], 'load_weights': 'True', 'batch_size': 128, 'device': 0}
net = build_model(os.path.join(ROOT, 'sub-cortical_segmentation_amd', 'nets'), opts)
n = 100000
ax, co, sa, at = synth.patch_batch(n, seed=0)
X = {'in1': ax, 'in2': co, 'in3': sa, 'in4': at}
net.predict_proba({k: v[:1000] for k, v in X.items()})
for _ in range(3):
    t = time.time()
    p = net.predict_proba(X)
    dt = time.time() - t
    print('predict_proba on host arrays: %.3f s -> %.0f patches/s (PCIe-inclusive)' % (dt, n / dt))
