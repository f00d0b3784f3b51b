#!/usr/bin/env python3
"""Reference-run fixtures for the HOST arithmetic that lives inside the reference's device-calling methods
(G15 ... G18): ``tests/golden/ref_goldens_host.npz``.

Test infrastructure; runs ONLY in the authoring container (the reference is not on the GPU box).  Data only is stored.

The reference's ``Demodulator`` (demodulator/demodulator_base.py) mixes numpy arithmetic with PyCUDA / cuFFT calls:
the Doppler table in ``__init__`` (DB:130-165), the pick interpolation around a ``memcpy_dtoh`` (DB:604-632), the
rate / phase arithmetic behind another (DB:733-752), the clamp, the symbol count and the float32 casts in front of the
``findCentres`` launch (DB:994-1006), the trust tagging and the return casts (DB:817-859).  None of that needs a GPU --
only somebody to stand where the driver stands.  This script installs a RECORDING FAKE of ``pycuda.driver`` /
``pycuda.compiler`` / ``lib.cufft``: allocations are sizes with a name, kernel launches and FFT executions are appended
to a log (name + scalar arguments), and every device-to-host copy hands out the bytes the harness queued for that
source buffer.  No arithmetic happens in the fake; everything recorded below was computed by the reference's own,
unmodified code -- constructor included -- on values injected at the exact points where its device results arrive.

  G15  Demodulator.__init__            Doppler tables, STX shift, rate window, clamp, thresholds      DB:75-240, 508-512
  G16  uploadAndFindCarrier (UHF)      injected findDopplerEst result + injected spectrum             DB:567-667
  G17  findCodeRateAndPhaseGPU +       injected {k*, arg, |P|^2}: spSym, codeOffset, and what the      DB:711-752, 991-1009
       cudaFindCentres                 findCentres launch gets (float32 casts, grid, symbol count)
  G18  demodulate (UHF / STX)          injected rate triple, symbols, centres, magnitudes, over        DB:765-859, 863-1051
                                       consecutive blocks: bits / centres / trust as returned
  G19  Demodulator_process.run         the caller's loop itself over its own SigFIFO (fake SUB socket,  DP:192-379, sigFIFO.py:108-181
                                       4095/4096-sample chunks) and Demodulator: block assembly, overlap
                                       carry, the result dict sent to the decoder

numpy's scalar promotion.  DB:623, DB:735 and DB:745 combine a float32 array element with a Python scalar.  The
reference uses ``np.float`` / ``np.int`` (DB:898, 1049; removed in numpy 1.24), i.e. it was written for and only runs
unshimmed on numpy < 1.24, where such an expression is evaluated in float64 ("legacy" value-based promotion).  Under
the numpy 2.2 of this image (NEP 50) the same text evaluates in float32.  Both readings are recorded:
  * ``nep50``  -- the reference as it runs here, untouched;
  * ``legacy`` -- the same code with the three-float read-back buffer ``__CodeRateAndPhaseResult`` allocated as float64
    (the injected values are float32-representable, so widening them is value-preserving and every later operation is
    then float64 -- exactly what numpy < 2 did at the first contact with a Python scalar).  ``bestDoppler`` is created
    inside ``__findUHF`` and cannot be widened; its only promotion-dependent use is ``sdev_Hz`` (DB:623), recorded as nep50.
The build follows ``legacy``.

Usage:  python tests/golden/make_golden_host.py
"""
import json
import logging
import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden as mg  # noqa: E402  (the shims of the first fixture set: np.float/np.int, crcmod, zmq)

REF = mg.REF


# ---- the recording fake ---------------------------------------------------------------------------------------------
class Buf:
    """A device allocation: a size.  ``int(buf)`` is its 'address'."""

    def __init__(self, nbytes, tag='mem'):
        self.nbytes, self.tag = int(nbytes), tag

    def __int__(self):
        return id(self)

    __index__ = __int__

    def free(self):
        pass


class Tape:
    log = []          # (name, scalar arguments ...)
    inject = {}       # source buffer -> array whose BYTES a memcpy_dtoh from it delivers
    by_tag = {}       # buffer tag -> deque of such arrays, one per copy (objects built where the harness cannot reach them: G19)


def _scalars(args):
    out = []
    for a in args:
        if isinstance(a, (int, float, np.integer, np.floating)):
            out.append(a)
        elif isinstance(a, tuple):
            out.append(tuple(a))
    return tuple(out)


class Kernel:
    def __init__(self, name):
        self.name = name

    def prepare(self, fmt):
        return self

    def prepared_call(self, grid, block, *args):
        Tape.log.append((self.name, tuple(grid), tuple(block)) + _scalars(args))


class _Pagelocked(np.ndarray):
    """What cuda.pagelocked_empty returns: an ndarray whose ``.base`` knows a device pointer (DB:457,460)."""

    @property
    def base(self):
        arr = self

        class _Alloc:
            def get_device_pointer(self):
                return arr.ctypes.data
        return _Alloc()


def _install_recording_fake():
    mg._install_shims()
    drv = sys.modules['pycuda.driver']
    comp = sys.modules['pycuda.compiler']
    fft = sys.modules['lib.cufft']

    class _Ctx:
        def pop(self):
            pass

    class Device:
        def __init__(self, idx):
            self.idx = idx

        def get_attribute(self, attr):
            return {'WARP_SIZE': 32, 'MANAGED_MEMORY': 1, 'MULTIPROCESSOR_COUNT': 80}[attr]

        def compute_capability(self):
            return (7, 0)

        def make_context(self):
            return _Ctx()

        def name(self):
            return 'recording fake'

    class SourceModule:
        def __init__(self, source, *a, **k):
            assert isinstance(source, str) and len(source) > 1000      # the reference read its own kernel file
        def get_function(self, name):
            return Kernel(name)

    def memcpy_dtoh(dest, src):
        val = Tape.inject[src] if src in Tape.inject else Tape.by_tag[src.tag].popleft()
        val = np.ascontiguousarray(val)
        dest.view(np.uint8).reshape(-1)[:] = val.view(np.uint8).reshape(-1)[:dest.nbytes]
        Tape.log.append(('memcpy_dtoh', src.tag, int(dest.nbytes)))

    def memcpy_htod(dest, src):
        Tape.log.append(('memcpy_htod', dest.tag, int(np.asarray(src).nbytes)))

    def pagelocked_empty(shape, dtype, mem_flags=0):
        return np.zeros(shape, dtype).view(_Pagelocked)

    drv.init = lambda: None
    drv.Device = Device
    drv.device_attribute = types.SimpleNamespace(WARP_SIZE='WARP_SIZE', MANAGED_MEMORY='MANAGED_MEMORY',
                                                 MULTIPROCESSOR_COUNT='MULTIPROCESSOR_COUNT')
    drv.host_alloc_flags = types.SimpleNamespace(DEVICEMAP=2)
    drv.mem_alloc = lambda n: Buf(n)
    drv.memcpy_dtoh = memcpy_dtoh
    drv.memcpy_htod = memcpy_htod
    drv.pagelocked_empty = pagelocked_empty
    drv.Context = types.SimpleNamespace(synchronize=lambda: None)
    comp.SourceModule = SourceModule
    fft.CUFFT_C2C, fft.CUFFT_R2C, fft.CUFFT_FORWARD, fft.CUFFT_INVERSE = 0x29, 0x2a, -1, 1
    fft.cufftPlan1d = lambda n, kind, batch: ('plan', int(n), int(kind), int(batch))
    fft.cufftDestroy = lambda plan: None
    fft.cufftExecC2C = lambda plan, i, o, d: Tape.log.append(('cufftExecC2C', plan[1], plan[3], int(d)))
    fft.cufftExecR2C = lambda plan, i, o: Tape.log.append(('cufftExecR2C', plan[1], plan[3]))
    logging.getLogger('pyCuSDR').setLevel(logging.CRITICAL + 1)


def _tag_buffers(obj):
    for k, v in vars(obj).items():
        if isinstance(v, Buf):
            v.tag = k


def build(backend, conf, proto, radio='UHF-H'):
    """The reference's constructor, whole, under the recording fake."""
    mod = __import__(f'demodulator.{backend}', fromlist=['Demodulator'])
    Tape.log = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        obj = mod.Demodulator(conf, proto, radio)
    _tag_buffers(obj)
    return obj


def ref_protocol(name, conf):
    if name == 'CC11xx':
        from protocol.CC11xx import CC11xx
        return CC11xx(conf=conf)
    m = __import__(f'protocol.benchmark.{name}', fromlist=['x'])
    return getattr(m, name[0].upper() + name[1:])(conf=conf)


# ---- G15 ------------------------------------------------------------------------------------------------------------
INIT_FIELDS = ('doppIdxNorm', 'doppHzLUT', 'doppCyperSymNorm', 'doppOffsetIdx', 'doppIdxArrayLen', 'doppIdxArrayOffset',
               'spsymMin', 'sampleRate', 'sigOverlapWin', 'windowWidthOffset', 'symbol_check_match_threshold', 'overlapOffset',
               'symbol_check_error_threshold', 'codeRateAndPhaseOffsetLow', 'codeRateAndPhaseOffsetHigh', 'num_masks',
               'Nfft', 'SUM_ALL_MASKS_PYTHON', 'CODE_SEARCH_MASK_OFFSET', 'num_dopplers', 'centreFreqOffset')


def g15_cases():
    from pycusdr_amd import config as cfg
    cases = {}
    for bs in (15, 16, 20):
        for D in (32, 64, 256, 1024, 2048):
            cases[f'bench_b{bs}_d{D}'] = ('UHF', 'bench_GMSK', cfg.bench_config('bench_GMSK', blockSize=bs, doppCarrierSteps=D))
    # the widened spans bench.py uses so that 1024 / 2048 shifts stay distinct at N = 2^20
    for D, rr in ((1024, 30000), (2048, 60000)):
        cases[f'bench_b20_d{D}_rr{rr}'] = ('UHF', 'bench_GMSK', cfg.bench_config('bench_GMSK', blockSize=20, doppCarrierSteps=D, rangeRateMax=rr))
    c = cfg.bench_config('bench_GMSK', blockSize=14, doppCarrierSteps=16)
    c['Radios']['Rx']['UHF-H']['noise_measure_offset_Hz'] = -38400
    cases['noise_neg'] = ('UHF', 'bench_GMSK', c)
    c = cfg.bench_config('bench_GMSK', blockSize=15, doppCarrierSteps=33)
    c['Radios']['Rx']['UHF-H']['noise_measure_offset_Hz'] = 19200.0
    cases['noise_pos'] = ('UHF', 'bench_GMSK', c)
    c = cfg.bench_config('bench_GMSK', blockSize=15, doppCarrierSteps=32)
    c['Radios']['Rx']['UHF-H']['frequencyOffset_Hz'] = 0
    cases['zero_if'] = ('UHF', 'bench_GMSK', c)              # the scan straddles 0 Hz: negative shifts wrap to the top
    c = cfg.bench_config('bench_GMSK', blockSize=15, doppCarrierSteps=64)
    c['Radios']['Rx']['UHF-H']['frequencyOffset_Hz'] = -38400
    cases['neg_if'] = ('UHF', 'bench_GMSK', c)
    cases['bpsk_b15'] = ('UHF', 'bench_BPSK', cfg.bench_config('bench_BPSK', blockSize=15, doppCarrierSteps=32))
    cases['cc11xx_b17_s128'] = ('UHF', 'CC11xx', cfg.cc11xx_config(blockSize=17, doppCarrierSteps=64, samplesPerSym=128))
    cases['cc11xx_b16_s16'] = ('UHF', 'CC11xx', cfg.cc11xx_config(blockSize=16, doppCarrierSteps=64, samplesPerSym=16))
    c = cfg.bench_config('bench_GMSK', blockSize=14, doppCarrierSteps=4)
    c['Radios']['Rx']['UHF-H']['CUDA_settings'] = 'STX'
    c['GPU']['STX'].update(blockSize=14, xcorrMaskSize=3)
    cases['stx_b14'] = ('STX', 'bench_GMSK', c)
    return cases


def g15_init(out):
    objs = {}
    for name, (backend, pname, conf) in g15_cases().items():
        obj = build(backend, conf, ref_protocol(pname, conf))
        out[f'g15/{name}/conf'] = np.array(json.dumps(conf))
        out[f'g15/{name}/backend'] = np.array(backend)
        out[f'g15/{name}/protocol'] = np.array(pname)
        for f in INIT_FIELDS:
            out[f'g15/{name}/{f}'] = np.asarray(getattr(obj, f))
        out[f'g15/{name}/doppOffsetIdx_dtype'] = np.array(type(obj.doppOffsetIdx).__name__)
        # what the constructor asked the device for: the cube the reference materialises (DB:436) and the launch shapes
        out[f'g15/{name}/xcorr_bytes'] = np.int64(obj.GPU_bufXcorr.nbytes)
        out[f'g15/{name}/centres_capacity'] = np.int64(obj.GPU_symbols.nbytes // 4)
        objs[name] = obj
    return objs


# ---- G16 ------------------------------------------------------------------------------------------------------------
def spectrum(N, seed):
    rs = np.random.RandomState(seed)
    X = (rs.standard_normal(N) + 1j * rs.standard_normal(N)).astype(np.complex64)
    X[N // 4 - 40:N // 4 + 40] *= 25
    X[:24] *= 9
    X[-24:] *= 9
    return X


def g16_pick(out, objs):
    for name in ('bench_b15_d64', 'zero_if', 'neg_if', 'noise_neg', 'noise_pos', 'cc11xx_b16_s16', 'cc11xx_b17_s128', 'bench_b20_d256'):
        obj = objs[name]
        N, Dtot = obj.Nfft, obj.doppIdxArrayLen
        seed = 1000 + sum(map(ord, name))
        obj.GPU_bufSignalFreq_cpu_handle[:] = spectrum(N, seed)
        out[f'g16/{name}/spectrum_seed'] = np.int64(seed)
        out[f'g16/{name}/spectrum_sha'] = np.array(mg.sha(np.asarray(obj.GPU_bufSignalFreq_cpu_handle)))
        picks = [0.0, 1.0, 5.0, 5.25, 7.5, 12.75, np.float32(14.333333), np.float32(30.999998), Dtot - 1, Dtot - 1.5,
                 Dtot - 2 + np.float32(1e-3), 0.5, np.nan]
        picks = [p for p in picks if not (p == p) or p <= Dtot - 1]
        metrics = np.linspace(-3.5, 41.0, len(picks)).astype(np.float32)
        rows = []
        for p, m in zip(picks, metrics):
            Tape.inject[obj.GPU_bufDoppResult] = np.array([p, m], dtype=np.float32)
            Tape.log = []
            samples = obj.get_signalBufferHostPointer()
            with warnings.catch_warnings(), np.errstate(all='ignore'):
                warnings.simplefilter('ignore')
                fo, sd, clipped, snr = obj.uploadAndFindCarrier(samples)
            launched = [e[0] for e in Tape.log]
            assert launched[:2] == ['cufftExecC2C', 'setArrayToZeros'] and 'findDopplerEst' in launched, launched
            est = [e for e in Tape.log if e[0] == 'findDopplerEst'][0]
            rows.append((fo, float(sd), float(snr), int(obj.dopplerIdxlast), int(est[3]), int(est[4]), type(sd).__name__,
                         type(obj.dopplerIdxlast).__name__))
        out[f'g16/{name}/pick'] = np.array(picks, dtype=np.float32)
        out[f'g16/{name}/metric'] = metrics
        out[f'g16/{name}/freqOffset'] = np.array([r[0] for r in rows], dtype=np.float64)
        out[f'g16/{name}/sdev_Hz_nep50'] = np.array([r[1] for r in rows], dtype=np.float64)
        out[f'g16/{name}/SNR'] = np.array([r[2] for r in rows], dtype=np.float64)
        out[f'g16/{name}/dopplerIdxlast'] = np.array([r[3] for r in rows], dtype=np.int64)
        out[f'g16/{name}/estDoppler_args'] = np.array([r[4:6] for r in rows], dtype=np.int64)
        out[f'g16/{name}/sdev_type_nep50'] = np.array(sorted({r[6] for r in rows}))
        out[f'g16/{name}/idx_type'] = np.array(sorted({r[7] for r in rows}))


# ---- G17 ------------------------------------------------------------------------------------------------------------
def _widen(obj, reading):
    """'legacy': the three-float read-back buffer as float64 (see the module docstring)."""
    key = '_Demodulator__CodeRateAndPhaseResult'
    setattr(obj, key, np.empty(3, dtype=np.float32 if reading == 'nep50' else np.float64))
    return getattr(obj, key).dtype


def _rate_and_centres(obj, reading, triple, capacity_arrays):
    """findCodeRateAndPhaseGPU + cudaFindCentres on an injected triple; what they return and what the launch got."""
    dt = _widen(obj, reading)
    Tape.inject[obj.GPU_bufCodeAndPhaseResult] = np.asarray(triple, dtype=np.float32).astype(dt)
    Tape.inject[obj.GPU_symbols], Tape.inject[obj.GPU_centres], Tape.inject[obj.GPU_magnitude] = capacity_arrays
    Tape.log = []
    with warnings.catch_warnings(), np.errstate(all='ignore'):
        warnings.simplefilter('ignore')
        spSym, off = obj.findCodeRateAndPhaseGPU()
        res = obj.cudaFindCentres(spSym, off)
    fc = [e for e in Tape.log if e[0] == 'findCentres'][0]          # (name, grid, block, spSymF, phaseF, N, op)
    return spSym, off, res, fc


def g17_rate(out, objs):
    """Dense sweep of the rate / phase arithmetic: every k* of the search window x a few phases."""
    import demodulator.demodulator_base as db
    for name in ('bench_b15_d64', 'bench_b20_d256', 'cc11xx_b17_s128'):
        obj = objs[name]
        N = obj.Nfft
        cap = int(N / obj.spsymMin)
        dummy = (np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.float32))
        ks = np.arange(obj.codeRateAndPhaseOffsetHigh, obj.codeRateAndPhaseOffsetLow)
        if len(ks) > 4000:
            ks = np.unique(np.concatenate((ks[:600], ks[-600:], ks[::7])))
        # also k* beyond the window: the clamp spSym < spsymMin (DB:994-995) and very few symbols
        ks = np.concatenate((ks, [N // obj.spsymMin, N // obj.spsymMin + 1, N // 2, N // 3, 5, 2, 1]))
        args = np.array([-3.1415925, -3.0, -1.5707964, -0.5, -1e-7, 0.0, 1e-7, 0.7, 1.5707964, 3.0, 3.1415927], dtype=np.float32)
        kk, aa = np.meshgrid(ks, args, indexing='ij')
        kk, aa = kk.ravel(), aa.ravel()
        for reading in ('nep50', 'legacy'):
            rows = np.empty((len(kk), 6), dtype=np.float64)
            for i, (k, a) in enumerate(zip(kk, aa)):
                spSym, off, res, fc = _rate_and_centres(obj, reading, (k, a, 1.0), dummy)
                rows[i] = (float(spSym), float(off), float(fc[3]), float(fc[4]), len(res[0]), fc[1][0])
            out[f'g17/{name}/{reading}/spSym'] = rows[:, 0]
            out[f'g17/{name}/{reading}/codeOffset'] = rows[:, 1]
            out[f'g17/{name}/{reading}/spSymF'] = rows[:, 2].astype(np.float32)
            out[f'g17/{name}/{reading}/phaseF'] = rows[:, 3].astype(np.float32)
            out[f'g17/{name}/{reading}/count'] = rows[:, 4].astype(np.int64)
            out[f'g17/{name}/{reading}/grid'] = rows[:, 5].astype(np.int64)
            out[f'g17/{name}/{reading}/spSym_type'] = np.array(type(spSym).__name__)
        out[f'g17/{name}/k'] = kk.astype(np.float32)
        out[f'g17/{name}/arg'] = aa.astype(np.float32)
        out[f'g17/{name}/op'] = np.int64(db.Operations.CENTRES_ABS.value)
    # k* = 0 (DB:737-740): what the reference really does
    obj = objs['bench_b15_d64']
    cap = int(obj.Nfft / obj.spsymMin)
    dummy = (np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.float32))
    for reading in ('nep50', 'legacy'):
        _widen(obj, reading)
        Tape.inject[obj.GPU_bufCodeAndPhaseResult] = np.array([0, 0.5, 0], dtype=getattr(obj, '_Demodulator__CodeRateAndPhaseResult').dtype)
        with warnings.catch_warnings(), np.errstate(all='ignore'):
            warnings.simplefilter('ignore')
            spSym, off = obj.findCodeRateAndPhaseGPU()
        out[f'g17/k_zero/{reading}/spSym'] = np.float64(spSym)
        out[f'g17/k_zero/{reading}/codeOffset'] = np.float64(off)


# ---- G18 ------------------------------------------------------------------------------------------------------------
def device_symbols(N, cap, M, spsym_nominal, block, ov, slip, seed):
    """What the three findCentres buffers of one block could hold: symbols of a continuous random stream seen through
    overlapping blocks (with an optional +-1 slip), centres on a jittered grid, positive float32 magnitudes."""
    rs = np.random.RandomState(seed)
    stream = np.random.RandomState(77).randint(0, M, 1 << 18).astype(np.int32)
    first = int(round(block * (N - ov) / spsym_nominal)) + slip
    sym = np.zeros(cap, np.int32)
    n = min(cap, len(stream) - first)
    sym[:n] = stream[first:first + n]
    x = np.arange(cap)
    cen = (x * np.float64(spsym_nominal) + 6 + rs.randint(-1, 2, cap)).astype(np.int64)
    cen = np.clip(cen, 0, N - 1).astype(np.int32)
    mag = (rs.gamma(2.0, 1e3, cap) + 1).astype(np.float32)
    return sym, cen, mag


def g18_demodulate(out, objs):
    scen = {
        'gmsk': dict(case='bench_b15_d64', slips=[0, 0, 1, 0, -1, 0], k_off=[0, 1, -2, 0, 3, -1], args=[0.3, -2.9, 1.2, 3.1, -0.01, 0.0]),
        'bpsk': dict(case='bpsk_b15', slips=[0, 1, 0, -1], k_off=[0, 0, 2, -1], args=[1.0, -1.0, 2.5, -0.2]),
        'cc11xx': dict(case='cc11xx_b17_s128', slips=[0, 0, -1, 0], k_off=[0, 1, 0, -1], args=[-3.0, 0.4, 2.2, -1.1]),
        'stx': dict(case='stx_b14', slips=[0, 1, 0], k_off=[0, 0, -1], args=[0.9, -0.6, 2.0]),
    }
    for sname, sc in scen.items():
        for reading in ('nep50', 'legacy'):
            backend, pname, conf = g15_cases()[sc['case']]
            obj = build(backend, conf, ref_protocol(pname, conf))     # fresh state (poswinP) per reading
            dt = _widen(obj, reading)
            N, M = obj.Nfft, obj.num_masks
            if obj.bitLUT is None:          # BPSK: the NRZ-S LUT has a row per sign-free pattern; strict '>' keeps the device
                M = len(obj.symbolLUT)      # on the lower of two sign-flipped filters (CU:130), i.e. below that
            ov = 2 ** obj.confGPU['overlap']
            cap = int(N / obj.spsymMin)
            k0 = int(round(N / obj.spsym))
            for b, (slip, dk, arg) in enumerate(zip(sc['slips'], sc['k_off'], sc['args'])):
                seed = 500 + 10 * b + sum(map(ord, sname))
                sym, cen, mag = device_symbols(N, cap, M, obj.spsym, b, ov, slip, seed)
                triple = np.array([k0 + dk, arg, 1e6], dtype=np.float32)
                Tape.inject[obj.GPU_bufCodeAndPhaseResult] = triple.astype(dt)
                Tape.inject[obj.GPU_symbols], Tape.inject[obj.GPU_centres], Tape.inject[obj.GPU_magnitude] = sym, cen, mag
                samples = obj.get_signalBufferHostPointer()
                if backend == 'STX':            # real samples: the clipping that feeds the trust tagging runs on them
                    rs = np.random.RandomState(seed)
                    x = (rs.standard_normal(N) + 1j * rs.standard_normal(N)).astype(np.complex64)
                    for pos in rs.randint(2000, N - 2000, 3):
                        x[pos:pos + 2] *= 80
                    samples[:] = x
                    out[f'g18/{sname}/b{b}/samples_seed'] = np.int64(seed)
                Tape.inject[obj.GPU_bufDoppResult] = np.array([3.5, 10.0], dtype=np.float32)
                Tape.log = []
                with warnings.catch_warnings(), np.errstate(all='ignore'):
                    warnings.simplefilter('ignore')
                    est = obj.uploadAndFindCarrier(samples)
                    bits, cw, tw, spSym = obj.demodulate()
                fc = [e for e in Tape.log if e[0] == 'findCentres'][0]
                sh = [e for e in Tape.log if e[0] == 'multInputVectorWithShiftedMask'][0]
                p = f'g18/{sname}/b{b}'
                if reading == 'nep50':
                    out[f'{p}/triple'] = triple
                    out[f'{p}/symbols'] = sym
                    out[f'{p}/centres_dev'] = cen
                    out[f'{p}/magnitudes'] = mag
                    out[f'{p}/clippedPeakIPure'] = np.asarray(obj.clippedPeakIPure, dtype=np.int64)
                    out[f'{p}/shift_arg'] = np.int64(sh[-1])      # (name, grid, block, spectrum pointer, shift)
                out[f'{p}/{reading}/bits'] = np.asarray(bits)
                out[f'{p}/{reading}/centres'] = np.asarray(cw)
                out[f'{p}/{reading}/trust'] = np.asarray(tw)
                out[f'{p}/{reading}/spSym'] = np.float64(spSym)
                out[f'{p}/{reading}/spSym_type'] = np.array(type(spSym).__name__)
                out[f'{p}/{reading}/spSymF'] = np.float32(fc[3])
                out[f'{p}/{reading}/phaseF'] = np.float32(fc[4])
                out[f'{p}/{reading}/out_dtypes'] = np.array([str(np.asarray(v).dtype) for v in (bits, cw, tw)])
            out[f'g18/{sname}/case'] = np.array(sc['case'])
            out[f'g18/{sname}/nblocks'] = np.int64(len(sc['slips']))



# ---- G19 ------------------------------------------------------------------------------------------------------------
def _install_fake_zmq(chunks, sent, on_empty):
    """The transport the reference's loop talks to, as far as the loop can tell: a SUB socket that delivers ``chunks`` (bytes) and
    then times out, a PUSH socket that keeps what is sent."""
    import copy
    z = sys.modules['zmq']
    z.SUB, z.PUSH, z.POLLIN, z.SUBSCRIBE, z.LINGER, z.NOBLOCK = 2, 8, 1, 6, 17, 1
    z.error = types.SimpleNamespace(Again=type('Again', (Exception,), {}))
    pending = list(chunks)

    class Socket:
        def connect(self, addr):
            pass

        def setsockopt_string(self, *a):
            pass

        def setsockopt(self, *a):
            pass

        def recv(self):
            return pending.pop(0)

        def send_pyobj(self, obj, flags=0):
            sent.append(copy.deepcopy(obj))

        def close(self):
            pass

    class Poller:
        def register(self, *a):
            pass

        def poll(self, timeout):
            if pending:
                return [(None, 1)]
            on_empty()
            return []

    z.Context = lambda: types.SimpleNamespace(socket=lambda kind: Socket())
    z.Poller = Poller


def g19_process_loop(out):
    """Demodulator_process.run (DP:192-357) itself: the reference's loop over its own SigFIFO (fed by a fake SUB socket with
    GNU-Radio-sized chunks) and its own Demodulator (under the recording fake, device results injected per block).  Pins the
    caller's side of the path: block assembly and overlap carry (hash of the buffer every block is uploaded from), the
    result dict that goes to the decoder -- keys, the constant entries, rangerate (computeTxFreqOffset DP:359-379),
    baudrate_est, the block counter."""
    import collections
    import demodulator_process as dp
    import demodulator.UHF as uhf
    from pycusdr_amd import config as cfg
    conf = cfg.bench_config('bench_GMSK', blockSize=13, doppCarrierSteps=16)
    conf['Demodulator'] = {'timeoutSeconds': 1}
    conf['Interfaces'] = {'Internal': {'demodOut': 'inproc://demod'}}
    conf['Radios']['Rx']['UHF-H']['RxInPort'] = 'inproc://rx'
    proto = ref_protocol('bench_GMSK', conf)
    N, ov, nblocks = 1 << 13, 1 << 10, 6
    rs = np.random.RandomState(19)
    stream = (rs.standard_normal(nblocks * (N - ov) + 777) + 1j * rs.standard_normal(nblocks * (N - ov) + 777)).astype(np.complex64)
    sizes, pos, chunks = [4095, 4096], 0, []
    while pos < len(stream):                       # GNU Radio hands over ~4095-4096 samples at a time (sigFIFO.py:160)
        n = sizes[len(chunks) % 2]
        chunks.append(stream[pos:pos + n].tobytes())
        pos += n
    cap = int(N / int(16 / 2))
    picks = [(3.5, 10.0), (7.25, 12.5), (7.0, 11.0), (np.nan, 0.0), (8.75, 9.0), (4.0, 14.0)]
    ks = [N // 16, N // 16 + 1, N // 16 - 1, N // 16, N // 16 + 2, N // 16]
    orig = uhf.Demodulator
    for reading in ('nep50', 'legacy'):            # see the module docstring: the read-back buffer as the reference has it / widened
        sent, raws = [], []
        dt = np.float32 if reading == 'nep50' else np.float64

        class Tapped(orig):
            def __init__(self, *a):
                orig.__init__(self, *a)
                _tag_buffers(self)
                _widen(self, reading)

            def uploadAndFindCarrier(self, samples):
                raws.append(mg.sha(np.asarray(samples)))
                return orig.uploadAndFindCarrier(self, samples)
        uhf.Demodulator = Tapped
        try:
            proc = dp.Demodulator_process(conf, proto, 'UHF-H')
            _install_fake_zmq(chunks, sent, on_empty=proc.runStatus.clear)
            Tape.by_tag = {k: collections.deque() for k in ('GPU_bufDoppResult', 'GPU_bufCodeAndPhaseResult', 'GPU_symbols', 'GPU_centres',
                                                             'GPU_magnitude')}
            for b in range(nblocks):
                sym, cen, mag = device_symbols(N, cap, 8, 16, b, ov, [0, 0, 1, 0, 0, -1][b], 1900 + b)
                triple = np.array([ks[b], 0.4 * b - 1.0, 1e6], dtype=np.float32)
                Tape.by_tag['GPU_bufDoppResult'].append(np.array(picks[b], dtype=np.float32))
                Tape.by_tag['GPU_bufCodeAndPhaseResult'].append(triple.astype(dt))
                Tape.by_tag['GPU_symbols'].append(sym)
                Tape.by_tag['GPU_centres'].append(cen)
                Tape.by_tag['GPU_magnitude'].append(mag)
                out[f'g19/b{b}/pick'] = np.array(picks[b], dtype=np.float32)
                out[f'g19/b{b}/triple'] = triple
                out[f'g19/b{b}/symbols'], out[f'g19/b{b}/centres_dev'], out[f'g19/b{b}/magnitudes'] = sym, cen, mag
            with warnings.catch_warnings(), np.errstate(all='ignore'):
                warnings.simplefilter('ignore')
                proc.run()
        finally:
            uhf.Demodulator = orig
            Tape.by_tag = {}
        assert len(sent) == len(raws) == nblocks, (len(sent), len(raws))
        out['g19/keys'] = np.array(sorted(sent[0].keys()))
        out['g19/raw_sha'] = np.array(raws)
        for k in ('workerId', 'voteGroup', 'baudRate', 'sample_rate', 'protocol', 'rangerateEst', 'baudRate_est'):
            assert all(d[k] == sent[0][k] for d in sent)
            out[f'g19/const/{k}'] = np.array(sent[0][k])
        for b, d in enumerate(sent):
            for k in ('count', 'doppler', 'doppler_std', 'SNR', 'rangerate'):       # independent of the reading
                out[f'g19/b{b}/{k}'] = np.float64(d[k])
            for k in ('spSymEst', 'baudrate_est'):
                out[f'g19/b{b}/{reading}/{k}'] = np.float64(d[k])
            out[f'g19/b{b}/{reading}/data'] = np.asarray(d['data'])
            out[f'g19/b{b}/{reading}/trust'] = np.asarray(d['trust'])
    out['g19/conf'] = np.array(json.dumps(conf))
    out['g19/stream_seed'] = np.int64(19)
    out['g19/stream_len'] = np.int64(len(stream))
    out['g19/chunk_sizes'] = np.array(sizes)
    out['g19/nblocks'] = np.int64(nblocks)


def main():
    if not os.path.isdir(REF):
        raise SystemExit('reference not mounted; fixtures can only be regenerated in the authoring container')
    _install_recording_fake()
    out = {'meta/numpy_version': np.array(np.__version__)}
    objs = g15_init(out)
    g16_pick(out, objs)
    g17_rate(out, objs)
    g18_demodulate(out, objs)
    g19_process_loop(out)
    flat = {k.replace('/', '__'): v for k, v in out.items()}
    path = os.path.join(HERE, 'ref_goldens_host.npz')
    np.savez_compressed(path, **flat)
    print(f'wrote {path}: {len(flat)} arrays, {os.path.getsize(path)/1e6:.2f} MB')


if __name__ == '__main__':
    main()
