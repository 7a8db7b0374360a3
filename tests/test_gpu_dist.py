"""The RCCL path on hardware: a ONE-rank `nccl` process group on cuda:0 with EVR_FORCE_DIST=1, so that every
collective of the drop-in (the checkpoint-load agreement, sequence_costs' broadcast, the per-dataset all-reduce of [total, count]:
eval.py:249-276,360-368 of the reference folded over ranks) is actually issued to RCCL.  A sum over one rank is the identity, so

  * `reduce_metric_sums` on a cuda float64 tensor must return MetricTracker's own totals for tests/golden/metric_tracker.json, and
  * `evaluate()` on the reference's own FireNet run (tests/golden/eval_loop.json: the reference's output files and dataset scores)
    must still reproduce those files and scores with the fold in the loop.

No scaling is measured here (one GPU); world-size-2 runs of the same code are tests/test_dist_cpu.py (gloo)."""
import os
import socket

import numpy as np
import pytest
import torch

from conftest import load_json

pytestmark = pytest.mark.gpu


@pytest.fixture()
def one_rank_rccl(monkeypatch):
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip('a process group already exists in this interpreter')
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    monkeypatch.setenv('MASTER_ADDR', '127.0.0.1'); monkeypatch.setenv('MASTER_PORT', str(port))
    monkeypatch.setenv('EVR_FORCE_DIST', '1')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
    try:
        yield dist
    finally:
        torch.cuda.synchronize()
        dist.destroy_process_group()


def test_all_reduce_of_metric_sums_on_rccl(one_rank_rccl):
    from evreal_amd.dist import reduce_metric_sums
    from evreal_amd.eval import fold_dataset_metrics
    from evreal_amd.eval_metrics import MetricTracker
    assert one_rank_rccl.get_backend() == 'nccl'
    for case in load_json('metric_tracker.json'):
        mt = MetricTracker()
        for key, value, count in case['updates']:
            mt.update(key, value, count)
        names = list(case['data'])
        sums = torch.zeros((len(names), 2), dtype=torch.float64, device='cuda')
        for i, nm in enumerate(names):
            sums[i, 0], sums[i, 1] = mt.data_dict[nm]['total'], mt.data_dict[nm]['count']
        tot = reduce_metric_sums(sums, one_rank_rccl)          # issues ncclAllReduce (EVR_FORCE_DIST)
        for i, nm in enumerate(names):
            assert tot[i, 0] == case['data'][nm]['total'] and int(tot[i, 1]) == case['data'][nm]['count'], (nm, tot[i])
        folded = fold_dataset_metrics(mt, names, one_rank_rccl)
        for nm in names:
            assert folded.get_count(nm) == case['data'][nm]['count'] and abs(folded.get_average(nm) - case['data'][nm]['average']) < 1e-15


def test_evaluate_under_a_one_rank_rccl_group(one_rank_rccl, tmp_path, monkeypatch):
    from test_gpu_eval import _evaluate_and_compare
    _evaluate_and_compare(tmp_path, monkeypatch, 2)      # two sequences per batch; files and dataset scores vs the reference's run
