"""
One-GPU box probe (round 4): what the pool allows for several processes on one card.
  1. a child process started (subprocess) by a parent that HAS initialised the GPU
  2. N gloo ranks, all on device 0: all_gather / send / recv with CUDA tensors
  3. N nccl ranks on device 0: RCCL's answer (expected: duplicate GPU refused)
Prints one JSON line per finding.
"""
import json
import os
import subprocess
import sys
import time

HERE = os.path.abspath(__file__)


def rank_main(backend: str) -> None:
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.cuda.set_device(0)
    out = {'backend': backend, 'rank': rank, 'world': world}
    try:
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', 0))
        else:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        x = torch.full((4,), float(rank + 1), device='cuda', dtype=torch.float64)
        try:
            dist.all_reduce(x)
            torch.cuda.synchronize()
            out['all_reduce_cuda'] = x.tolist()
        except Exception as e:  # noqa: BLE001
            out['all_reduce_cuda_error'] = f'{type(e).__name__}: {e}'[:400]
        try:
            g = [torch.zeros(4, device='cuda', dtype=torch.float64) for _ in range(world)]
            w = dist.all_gather(g, torch.full((4,), float(rank + 1), device='cuda', dtype=torch.float64), async_op=True)
            w.wait()
            torch.cuda.synchronize()
            out['all_gather_cuda'] = [t[0].item() for t in g]
        except Exception as e:  # noqa: BLE001
            out['all_gather_cuda_error'] = f'{type(e).__name__}: {e}'[:400]
        try:
            ops = []
            recv = [torch.zeros(4, device='cuda', dtype=torch.float64) for _ in range(world)]
            mine = torch.full((4,), float(rank + 1), device='cuda', dtype=torch.float64)
            for p in range(world):
                if p != rank:
                    ops.append(dist.P2POp(dist.isend, mine, p))
                    ops.append(dist.P2POp(dist.irecv, recv[p], p))
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            torch.cuda.synchronize()
            out['p2p_cuda'] = [t[0].item() for t in recv]
        except Exception as e:  # noqa: BLE001
            out['p2p_cuda_error'] = f'{type(e).__name__}: {e}'[:400]
    except Exception as e:  # noqa: BLE001
        out['init_error'] = f'{type(e).__name__}: {e}'[:800]
    print(json.dumps(out), flush=True)
    try:
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        pass


def launch(backend: str, world: int, timeout: float = 240.0) -> None:
    import socket

    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY='0', NCCL_DEBUG='WARN')
        procs.append(subprocess.Popen([sys.executable, HERE, 'rank', backend], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    t0 = time.time()
    for r, p in enumerate(procs):
        try:
            o, _ = p.communicate(timeout=max(5.0, timeout - (time.time() - t0)))
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
            o += '\n[probe] killed after timeout'
        print(json.dumps({'launch': backend, 'world': world, 'rank': r, 'rc': p.returncode, 'tail': o[-1500:]}), flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'rank':
        rank_main(sys.argv[2])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'child':
        import torch

        print(json.dumps({'child_of_gpu_parent': float(torch.ones(3, device='cuda').sum().item())}), flush=True)
        sys.exit(0)
    which = sys.argv[1:] or ['gloo2', 'gloo4', 'nccl2', 'child']
    # the multi-rank launches come first: this parent has not touched the GPU yet
    if 'gloo2' in which:
        launch('gloo', 2)
    if 'gloo4' in which:
        launch('gloo', 4)
    if 'nccl2' in which:
        launch('nccl', 2, timeout=120.0)
    if 'child' in which:
        import torch

        print(json.dumps({'parent_gpu': float(torch.ones(3, device='cuda').sum().item())}), flush=True)
        p = subprocess.run([sys.executable, HERE, 'child'], capture_output=True, text=True, timeout=300)
        print(json.dumps({'child_rc': p.returncode, 'child_out': (p.stdout + p.stderr)[-800:]}), flush=True)
