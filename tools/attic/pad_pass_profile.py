import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import torch
from mpi4py_fft_amd import PFFT, comm, _lib
def run(shape, dt, **kw):
    fft = PFFT(comm.COMM_SELF, shape, dtype=dt, **kw)
    t = fft.forward.input_array.tensor
    (torch.view_as_real(t) if t.is_complex() else t).view(-1)[: 1 << 28].normal_()
    for _ in range(2):
        fft.forward(); fft.backward()
    _lib.set_option('profile', 1)
    for _ in range(5):
        fft.forward(); fft.backward()
    torch.cuda.synchronize()
    _lib.set_option('profile', 0)
    plans = list(fft._fused_plans) if fft._fused_plans else [x.fwd for x in fft.xfftn] + [x.bck for x in fft.xfftn]
    print('%s %s %s' % (shape, dt, kw))
    for name, p in zip(('fwd', 'bwd') if fft._fused_plans else ['f0','f1','f2','b0','b1','b2'], plans):
        print(p._eng.plan_describe(p._plan) if hasattr(p,'_eng') else '')
        for fam, nbytes, ms, n in p.profile():
            if n:
                print('   %s %-20s %8.3f ms  %7.1f GB/s' % (name, fam, ms / n, nbytes / (ms / n) / 1e6))
    fft.destroy()
run((1024,512,512), 'f', padding=[1.5]*3)
run((1024,512,512), 'd', padding=[1.5]*3)
run((1024,1024,1024), 'f', padding=[1.5]*3)
run((512,512,512), 'f', padding=[1.5]*3)
