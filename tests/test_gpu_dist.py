"""The multi-GPU plumbing on the GPU box (which has ONE GPU): (1) the work-distribution collectives on the nccl (= RCCL) backend with
device tensors, world size 1 -- broadcast, all-reduce, gather and barrier all go through RCCL's code path; (2) `python bench.py --gpus 2`
on its own starts two ranks (child processes, launched before anything touched the GPU) and prints a line with n_gpus = 2 (both
ranks share the one device; gloo carries the collectives because RCCL refuses two ranks on one device).
Each case runs in a child process: a process group is process-wide state and must not leak into the other GPU tests."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

NCCL_CHILD = r'''
import os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
import __graft_entry__ as g
pkg = g.load_package()
D = __import__('importlib').import_module(pkg.__name__ + '.distribute')
torch.cuda.set_device(0)
dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%(port)d', rank=0, world_size=1, device_id=torch.device('cuda', 0))
dev = torch.device('cuda', 0)
dd = D.Distributor(dist, dev)
assert dd.world == 1 and dd.rank == 0
table = dd.broadcast_object([dict(id=3, modcod=14, weight=2.5)])
assert table == [dict(id=3, modcod=14, weight=2.5)]
t = torch.arange(64, dtype=torch.uint8, device=dev)
assert dd.broadcast_tensor(t).tolist() == list(range(64))
payload = torch.tensor([[5, 6, 7], [8, 9, 10]], dtype=torch.uint8, device=dev)
out, cnt = dd.gather_units([2, 0], payload, torch.tensor([3, 1], dtype=torch.int32, device=dev), 4)
assert out.is_cuda and out.tolist() == [[8, 9, 10], [0, 0, 0], [5, 6, 7], [0, 0, 0]] and cnt.tolist() == [1, 0, 3, 0]
assert dd.max_over_ranks(1.25) == 1.25
dd.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print('NCCL_OK', torch.cuda.nccl.version() if hasattr(torch.cuda, 'nccl') else '')
'''


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.gpu
def test_distribution_collectives_on_rccl_world_size_one():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-c', NCCL_CHILD % dict(root=ROOT, port=_free_port())], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'NCCL_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.gpu
def test_bench_gpus_2_starts_two_ranks_by_itself():
    env = dict(os.environ, DVBS2GPU_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--streams', '128', '--frames', '2',
                        '--no-secondary', '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    line = lines[0]
    assert line['n_gpus'] == 2 and line['scaling'] == 'weak' and line['value'] > 0
    assert line['config']['frames_delivered_checked'] > 0


@pytest.mark.gpu
def test_config4_sharded_over_two_ranks_delivers_what_one_rank_delivers():
    """BASELINE config 4 (independent transponders, main.cpp:588,595) on the real engine with TWO ranks: table + configuration broadcast from rank 0, the MODCOD-grouped
    weighted assignment, every rank's receivers, the per-step gather of BBFRAMEs + byte counts to the egress rank -- and what the egress rank holds, in transponder
    order, must be byte for byte what ONE rank delivers for the same table (both ranks share the box's one device; gloo carries the collectives)"""
    def run(gpus):
        env = dict(os.environ, DVBS2GPU_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
        for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
            env.pop(k, None)
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--config', 'mixed64', '--gpus', str(gpus), '--mixed-sub', '2', '--mixed-frames', '1',
                            '--steps', '2', '--warmup', '1'], env=env, capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith('{')]
        assert len(lines) == 1, r.stdout[-2000:]
        return lines[0]
    two, one = run(2), run(1)
    m2, m1 = two['mixed64'], one['mixed64']
    assert two['n_gpus'] == 2 and m2['n_gpus'] == 2 and m2['scaling'] == 'strong'
    assert len(m2['transponders_per_rank']) == 2 and min(m2['transponders_per_rank']) > 0 and sum(m2['transponders_per_rank']) == 64
    # (frames that are not transmitted ones: carriers whose loops had not settled or slipped at these Es/N0 -- the same ones whatever the number of ranks)
    assert m2['frames_not_transmitted_ones'] == m1['frames_not_transmitted_ones'] <= m1['frames_at_egress_last_step'] // 16
    assert m2['frames_at_egress_last_step'] == m1['frames_at_egress_last_step'] >= 100
    assert m2['egress_sha256'] == m1['egress_sha256'] and m2['egress_sha256']
