"""Worker of tests/test_gpu_train.py::test_data_parallel_fit_* (launched with torch.distributed.run): every rank runs
Net.fit on the same toy problem; rank r takes mini-batch s*world + r of each step, the hold-out pass is sharded."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'sub-cortical_segmentation_amd')):
    sys.path.insert(0, p)
from cnn_cort import dist, synth  # noqa: E402
from cnn_cort.nets import Net  # noqa: E402


def main(out_dir):
    rank, world, local = dist.init_from_env()
    torch.cuda.set_device(local)
    n = 7 * 32 + 9                                   # 7 full mini-batches + a ragged one over 2 ranks: 4 lock-step steps
    ax, co, sa, at = synth.patch_batch(n + 80, seed=21)
    y = at.argmax(1).astype(np.uint8)
    at = (0.1 * at + 0.9 * np.eye(15, dtype=np.float32)[y]).astype(np.float32)
    net = Net(max_epochs=2, patience=20, eval_size=0.25, batch_size=32, verbose=0, device=local, seed=2)
    net.fit({'in1': ax, 'in2': co, 'in3': sa, 'in4': at}, y)
    params = net.get_all_params_values()
    flat = np.concatenate([np.asarray(a, np.float32).ravel() for k in sorted(params) for a in params[k]])
    # the hold-out loss of the final parameters over the WHOLE validation set, computed by this rank alone
    from cnn_cort.nets import stratified_first_fold
    tr, va = stratified_first_fold(y, 4)
    h = net._ensure_handle()
    dev = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to('cuda:%d' % local, dtype=dt)
    ls, nc = h.eval_batch(dev(ax[va]), dev(co[va]), dev(sa[va]), dev(at[va]), dev(y[va], torch.uint8), len(va),
                          stream=torch.cuda.current_stream().cuda_stream)
    np.savez(os.path.join(out_dir, 'fit_rank%d.npz' % rank), params=flat, whole_valid_loss=ls / len(va), whole_valid_acc=nc / len(va),
             backend=str(torch.distributed.get_backend()))
    with open(os.path.join(out_dir, 'fit_rank%d.json' % rank), 'w') as f:
        json.dump(net.train_history_, f, default=float)
    dist.barrier()


if __name__ == '__main__':
    main(sys.argv[1])
