"""Probe: can two RCCL ranks share one GPU on this box?  (torchrun, backend nccl, both on cuda:0)
Also prints host facts the CPU baseline needs (cores, memory, libfftw3 presence)."""
import ctypes.util
import os
import sys
import torch
import torch.distributed as dist


def main():
    rank = int(os.environ.get('RANK', '0'))
    if rank == 0:
        print('cores', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
        with open('/proc/meminfo') as f:
            print(''.join(f.readlines()[:3]))
        for name in ('fftw3', 'fftw3_threads', 'fftw3_omp', 'fftw3f', 'mkl_rt'):
            print('find_library', name, ctypes.util.find_library(name))
        print('gpus', torch.cuda.device_count(), flush=True)
    torch.cuda.set_device(0)
    dist.init_process_group('nccl')
    n = dist.get_world_size()
    x = torch.full((n * 4,), float(rank), device='cuda')
    y = torch.empty_like(x)
    dist.all_to_all_single(y, x)
    torch.cuda.synchronize()
    print('rank', rank, 'alltoall ok', y.tolist(), flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
