"""Host CPU share of this process and the BLAS worker pool.

numpy's OpenBLAS starts one worker per core it sees (64 on the 256-core GPU hosts) and its idle workers spin for a
while after every call.  Inside a container with a CPU quota (the GPU boxes: 16 CPUs per 100 ms, cpu.max "1600000
100000") a single ``np.linalg.norm`` in the stimulus generator can spend the whole quota in those spinning workers,
and the kernel then stops EVERY thread of the process -- the receive loop included -- for the rest of the 100 ms period.
Measured with tools/ber_rows.py (profiles/r03_ber.md): one block of 70-83 ms among blocks of 0.13 ms, thread CPU time
across it 0.4 ms, cgroup nr_throttled 2 -> 50; gone with one BLAS thread.  The receive chain itself does not use BLAS
(its arithmetic is on the device), so the entry scripts call :func:`quiet_blas` once at start."""
import os


def cpu_share():
    """CPUs this process may use: the smaller of its affinity mask and its cgroup quota (v2 cpu.max or v1 cfs_quota)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f:
                quota = int(f.read())
            with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f:
                period = int(f.read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return n


def quiet_blas(threads=1):
    """Limit the BLAS pools already loaded in this process to ``threads`` workers (threadpoolctl), and set the
    environment defaults for pools loaded later and for child processes.  Returns the number of pools changed."""
    try:
        threads = int(os.environ.get('OPENBLAS_NUM_THREADS', threads))        # an explicit setting wins
    except ValueError:
        pass
    for var in ('OPENBLAS_NUM_THREADS', 'OMP_NUM_THREADS', 'MKL_NUM_THREADS'):
        os.environ.setdefault(var, str(threads))
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
    except ImportError:
        return 0
    threadpool_limits(limits=threads, user_api='blas')
    return sum(1 for p in threadpool_info() if p.get('user_api') == 'blas')
