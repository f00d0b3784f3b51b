"""Where does the sharded exchange spend its extra time (1-rank rehearsal)?"""
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29544', RANK='0', WORLD_SIZE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
from pycusdr_amd.mfbank import MFBank
log2N, D, M = 20, 256, 8
N = 1 << log2N
rs = np.random.RandomState(0)
x = (rs.standard_normal(N) + 1j * rs.standard_normal(N)).astype(np.complex64)
masks = (rs.standard_normal((M, N)) + 1j * rs.standard_normal((M, N))).astype(np.complex64)
bank = MFBank(log2N, D, M); bank.set_filters(masks); bank.set_shifts(np.arange(D) * 4099 % N); bank.upload(x)
st = torch.cuda.Stream(); scores = torch.zeros((D, M), device='cuda'); torch.cuda.synchronize()
bank.set_stream(st.cuda_stream)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def plain(): bank.find_carrier()
def zero_export():
    with torch.cuda.stream(st):
        scores.zero_(); bank.search_async(); bank.export_scores_async(scores.data_ptr(), 0); bank.pick(scores.data_ptr(), num=D, offset=0)
def full():
    with torch.cuda.stream(st):
        scores.zero_(); bank.search_async(); bank.export_scores_async(scores.data_ptr(), 0)
        dist.all_reduce(scores); bank.pick(scores.data_ptr(), num=D, offset=0)
def only_ar():
    with torch.cuda.stream(st):
        dist.all_reduce(scores)
    st.synchronize()
print(f'plain {t(plain):.3f} ms | +zero/export {t(zero_export):.3f} ms | +all_reduce {t(full):.3f} ms | all_reduce alone (sync) {t(only_ar, 200)*1e3:.1f} us')
dist.destroy_process_group()
