#!/usr/bin/env python3
"""bench.py -- LC3 frames/sec @ 48 kHz / 10 ms on 1..8 MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic PCM already resident in HBM.

  --mode roundtrip (default, BASELINE configs[1]): encode S streams x T frames (150-byte frames) and decode the
      bitstream that was just produced; S*T = 65 536 frames per GPU, stream state carried from step to step.
      With --gpus N every rank owns its own S streams ("scaling": "weak").
  --mode encode --frames-total 1048576 (BASELINE configs[2]): encode-only; the batch's streams are sharded
      contiguously over the ranks (lc3-codec_amd/dist.py::shard_range), "scaling": "strong".

Multi-GPU: `python bench.py --gpus N` starts N ranks itself (one process per GPU: RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* set per child, rendezvous on 127.0.0.1) BEFORE anything touches the GPU -- the parent never imports torch,
relays rank 0's JSON line and exits non-zero if a rank fails.  Launched under torch.distributed.run (WORLD_SIZE
already set) it is simply one rank.  Frames are independent across streams: there is no data-path collective;
torch.distributed (RCCL) carries the barrier and one all_reduce of {max elapsed, frames, parity mismatches}.

Prints ONE JSON line on rank 0 (DESIGN.md section "Measurement").  PyTorch is used for device memory, the HIP
stream, events and torch.distributed only; the codec is liblc3gpu.so (hand-written HIP) called through its C ABI.
`--engine emu` (tests only) swaps the GPU engine for the CPU wave emulator of the device code and RCCL for gloo, so
that the launcher and the sharding / reduction path run in a GPU-less container (tests/test_dist_gloo.py).
"""
import argparse
import hashlib
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FS, US, NBYTES, NF = 48000, 10000, 150, 480
ALG_BYTES_ENC = 2 * NF + NBYTES   # i16 PCM read + frame bytes written   
ALG_BYTES_DEC = NBYTES + 2 * NF   # frame bytes read + i16 PCM written
HBM_PEAK_GBS = 8000.0             # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
CLOCK_MHZ = 2400.0                # max shader clock (MI355X_MICROARCH.md, chip-level parameters)
N_SIMD = 1024                     # 256 CUs x 4 SIMDs
KERNEL_EVENTS_EVERY = 10          # the library's per-kernel HIP events are recorded on every tenth step of the timed region (16 records a step: ~2 % of it)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100,
                    help="timed steps (default 100: ~0.11 s, the resident 64 frames of every stream walked through six times; 20 until round 5 -- "
                         "a 23 ms region, a tenth of which is the chip filling and draining)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=("roundtrip", "encode"), default="roundtrip")
    ap.add_argument("--streams", type=int, default=None, help="streams per GPU (roundtrip; default 16384)")
    ap.add_argument("--frames", type=int, default=None, help="consecutive frames per stream per step (default 4; 16 in encode mode)")
    ap.add_argument("--frames-total", type=int, default=None,
                    help="encode mode: frames per step over ALL ranks (default 1048576 = BASELINE configs[2])")
    ap.add_argument("--hip-streams", type=int, default=1,
                    help="split the rank's streams over this many codec handle pairs, each on its own HIP stream "
                         "(default 1: one stream, clean per-kernel timing)")
    ap.add_argument("--no-overlap-probe", action="store_true",
                    help="skip timing the OTHER arrangements (see --arrangement) after the timed region")
    ap.add_argument("--arrangement", type=arrangement_name, default=arrangement_name("pipeline"),
                    help="how the timed steps are queued: `pipeline` (default) = the library's pipeline object (lc3gpu_pipeline_submit: two groups of the "
                         "streams, each with an encoder handle on a high-priority HIP stream and a decoder handle on a stream of its own -- the `quad` "
                         "arrangement below as a feature of the C ABI); `single` = encode then decode of the same batch on ONE HIP stream (every call waits for "
                         "the one before it); `pipelined` = the recommended caller pattern (INTEGRATION.md): the encoder handle on one HIP stream, the "
                         "decoder handle on another, two byte buffers and events, so that the decoder works on step k while the encoder runs step k + 1; "
                         "`staggered` = the same with three buffers and the decoder call queued behind the encoder's LC3GPU_ENC_STAGE_BACK event of the "
                         "next step (lc3gpu_encoder_stage_event); `split:a+b[+c..]` = the streams in as many equal groups as there are numbers, every group with "
                         "an encoder handle and a decoder handle of its own, on 2 HIP streams (`pipelined` by itself) or on 1 (encode then decode): "
                         "`quad` = split:2+2, `tri` = split:2+1, `duo` = split:1+1.  The other arrangements are timed too and reported beside `value`")
    ap.add_argument("--also", type=arrangement_name, action="append", default=None,
                    help="further arrangements to time after the timed region, beside single / pipelined / staggered (repeatable; default: none)")
    ap.add_argument("--sustain-seconds", type=float, default=2.5,
                    help="length of the sustained leg: back-to-back steps for this long, frames/s and the shader clock read by a one-wave probe kernel "
                         "beside them (0 = skip)")
    ap.add_argument("--resident-frames", type=int, default=None,
                    help="frames per stream resident in HBM (roundtrip mode; default 64 = 1 GB of PCM): step k codes frames [T k, T k + T) of "
                         "every stream, wrapping around, so that the timed region is not one loop over the same T frames")
    ap.add_argument("--no-other-modes", action="store_true",
                    help="skip SURVEY 8d's other shapes of the 65 536-frame batch (65 536 x 1 cold / carried, 4 096 x 16, the T sweep) and the "
                         "host-resident leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--engine", choices=("gpu", "emu"), default="gpu", help=argparse.SUPPRESS)
    a = ap.parse_args(argv)
    if a.also is None:
        a.also = []  # (`--also quad`: the hand-built form of `pipeline`; its streams are created AFTER the pipeline's, and a process's second set of
        #  four streams shares hardware queues -- 52 M where the same arrangement created first measures 58.6: profiles/r06_pipeline_stream_order.txt)
    return a


# ---------------------------------------------------------------------------------------------------------------
# launcher: N ranks as child processes, started before any GPU call
# ---------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv, limit_s=None):
    """Start n ranks of this script (one per GPU) as FRESH child processes, relay rank 0's JSON line; -> process exit code.
    All children are watched while they run: as soon as one exits non-zero (a rank without a GPU, an import error, a crash after
    the rendezvous) the remaining children -- exactly the processes started here -- are terminated and the launcher returns 1;
    otherwise rank 0 would sit in a collective until the backend's timeout.  An overall wall-clock limit covers hangs."""
    import tempfile

    port = _free_port()
    limit_s = float(os.environ.get("LC3_BENCH_LIMIT_S", "1800")) if limit_s is None else limit_s
    procs = []
    out0 = tempfile.TemporaryFile()  # rank 0's stdout (a file, not a pipe: nobody has to drain it while we poll)
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), LC3_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    deadline, rc, why = time.time() + limit_s, 0, None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            rc, why = 1, "rank %d exited with code %d" % bad[0]
            break
        if all(c == 0 for c in codes):
            break
        if time.time() > deadline:
            rc, why = 1, "wall-clock limit of %.0f s reached" % limit_s
            break
        time.sleep(0.05)
    if rc:
        for p in procs:  # the children this launcher started, nothing else
            if p.poll() is None:
                p.terminate()
        t_kill = time.time() + 10.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_kill - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        sys.stderr.write("bench.py: %s; the other ranks were stopped\n" % why)
        return 1
    out0.seek(0)
    line = None
    for ln in out0.read().decode(errors="replace").splitlines():
        if ln.startswith("{"):
            line = ln
    if line is None:
        sys.stderr.write("bench.py: rank 0 printed no JSON line\n")
        return 1
    print(line, flush=True)
    return 0


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline (oracle, test infrastructure): 1 thread and all host cores, persistent codec objects
# ---------------------------------------------------------------------------------------------------------------
def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def granted_cpus():
    """-> (threads this process may run on, how that was determined).  os.cpu_count() is the machine; what the job is granted is
    the affinity mask capped by the cgroup CPU quota (cpu.max / cfs_quota_us)."""
    try:
        aff = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        aff = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota|max> <period>"
            q, per = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = float(f.read()), float(g.read())
                if q > 0:
                    quota = q / per
        except (OSError, ValueError):
            pass
    n = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    return n, {"os_cpu_count": os.cpu_count(), "sched_getaffinity": aff, "cgroup_cpu_quota": quota}


def cpu_baseline(mode, seconds=2.5):
    """Time the CPU oracle (a C port of the reference; the reference itself is Rust and cannot be built on the box) on a bounded
    sample of the bench workload.  One C call per leg (oracle/lc3_batch.c::lc3o_timed_run): N threads, each with ONE persistent
    encoder / decoder object, code 128-frame streams of the bench generator's PCM for `seconds`; thread creation, allocation and
    initialisation lie outside the timed region.  Legs: 1, 2, 4, ... threads up to what the job is granted (affinity mask capped by
    the cgroup quota -- not os.cpu_count())."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    synth = importlib.import_module("lc3-codec_amd.synth")
    granted, how = granted_cpus()
    T = 128
    base = synth.make_pcm(64, 8, NF, FS)                 # 64 distinct streams of the bench generator ...
    pcm = np.ascontiguousarray(np.tile(base, (1, T // 8, 1)))  # ... 128 consecutive frames each
    legs, n = [], 1
    while n < granted:
        legs.append(n)
        n *= 2
    legs.append(granted)
    scaling = []
    for th in legs:
        frames, dt = O.timed_run(pcm, NBYTES, FS, US, threads=th, roundtrip=(mode == "roundtrip"), seconds=seconds)
        scaling.append({"threads": th, "value": frames / dt, "frames": int(frames), "seconds": dt})
    v1, va = scaling[0], scaling[-1]
    # second leg: the same sources as a user's release build would compile them -- -O3 and this host's instruction set, the strict-IEEE flags
    # kept (oracle/Makefile `native`, built HERE: -march=native code does not travel) -- checked bit-identical on a sample before it is timed
    native = None
    Ln = O.lib_native()
    if Ln is not None:
        sample = np.ascontiguousarray(pcm[:16, :16])
        b0 = O.encode_batch(sample, NBYTES, FS, US)
        b1 = O.encode_batch(sample, NBYTES, FS, US, library=Ln)
        same = bool(np.array_equal(b0, b1) and np.array_equal(O.decode_batch(b0, NF, FS, US), O.decode_batch(b0, NF, FS, US, library=Ln)))
        n1 = O.timed_run(pcm, NBYTES, FS, US, threads=1, roundtrip=(mode == "roundtrip"), seconds=seconds, library=Ln)
        na = O.timed_run(pcm, NBYTES, FS, US, threads=granted, roundtrip=(mode == "roundtrip"), seconds=seconds, library=Ln)
        native = {"flags": "-O3 -march=native -ffp-contract=off -fno-fast-math", "bit_identical_to_the_default_build": same,
                  "threads_1": n1[0] / n1[1], "value": na[0] / na[1], "cores": granted, "unit": "frames/s"}
    what = "encode+decode" if mode == "roundtrip" else "encode"
    eff = va["value"] / (v1["value"] * va["threads"])
    note = None
    if eff < 0.5:
        note = (f"all-threads / 1-thread = {va['value'] / v1['value']:.1f}x on {va['threads']} threads (efficiency {eff:.2f}): the threads are "
                "SMT siblings / share memory bandwidth and boost clocks; see `scaling` for where it flattens")
    return {
        "value": va["value"], "unit": "frames/s", "cores": va["threads"], "kind": "port",
        "threads_1": {"value": v1["value"], "unit": "frames/s", "cores": 1,
                      "sample": f"{v1['frames']} frames, {what}, {v1['seconds']:.1f} s"},
        "scaling": scaling, "parallel_efficiency": eff, "granted": how, "cpu_model": _cpu_model(),
        "build": "`value` and `scaling` are the default oracle build: gcc -O2, no -march (conservative flags, the build the parity tests use); "
                 "`native_build` is the same source with -O3 -march=native on this host",
        "native_build": native,
        "sample": f"{va['frames']} frames of the bench generator's PCM (64 distinct streams x {T} consecutive frames, one persistent "
                  f"codec object per thread, a stream per pass), {what}, {va['threads']} host threads, {va['seconds']:.1f} s; "
                  "threads, buffers and codec objects are created before the clock starts",
        "note": note,
    }


# ---------------------------------------------------------------------------------------------------------------
# committed PMC summary (rocprofv3 cannot run inside this process): only trusted for the kernel sources it measured
# ---------------------------------------------------------------------------------------------------------------
def kernel_source_sha():
    """sha256 over the device code (csrc/*, tables): stamps profiles/pmc_latest.json (tools/pmc_json.py)"""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "lc3-codec_amd", "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".h", ".hip"))) + [os.path.join(ROOT, "tables", "lc3_tables.h")]
    for p in files:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


SPLIT_ALIASES = {"quad": "split:2+2", "tri": "split:2+1", "duo": "split:1+1"}


def arrangement_name(a):
    """argparse type of --arrangement: single | pipelined | staggered | split:N+N.. (N = 1 or 2 HIP streams of a group) | quad | tri | duo"""
    a = SPLIT_ALIASES.get(a, a)
    if a in ("single", "pipelined", "staggered", "pipeline"):
        return a
    if a.startswith("split:") and all(x in ("1", "2") for x in a[6:].split("+")) and 2 <= len(a[6:].split("+")) <= 4:
        return a
    raise argparse.ArgumentTypeError("arrangement: pipeline, single, pipelined, staggered, quad, tri, duo or split:a+b[+c[+d]] with a, b .. in {1, 2}")


def arrangement_streams(a):
    if a == "pipeline":
        return 4
    return sum(int(x) for x in a[6:].split("+")) if a.startswith("split:") else (1 if a == "single" else 2)


def load_ceiling():
    """the measured issue ceiling and copy bandwidth (tools/valu_ceiling.hip -> profiles/r04_valu_ceiling.json): cycles per wave64
    vector instruction of the codec's instruction mix on one SIMD, by waves per SIMD, and the device copy rate"""
    try:
        with open(os.path.join(ROOT, "profiles", "r04_valu_ceiling.json")) as f:
            j = json.load(f)
        mix = {r["waves_per_simd"]: r["cycles_per_wave_instr"] for r in j["valu"] if r["instruction"].startswith("mix")}
        return {"cycles_per_instr_by_waves_per_simd": mix, "copy_GBs": j["copy"]["GBs_read_plus_write"],
                "source": "profiles/r04_valu_ceiling.json (tools/valu_ceiling.hip on an MI355X of this pool)"}
    except (OSError, ValueError, KeyError):
        return None


# resident waves per SIMD of each kernel on the 65 536-frame batch (lane-per-frame kernels: 1 024 waves on 1 024 SIMDs, or 1 024 wave PAIRS)
WAVES_PER_SIMD = {"lc3_enc_front_kernel": 4, "lc3_sns_vq_kernel": 1, "lc3_enc_back_kernel": 4, "lc3_pack_kernel": 2,
                  "lc3_parse_kernel": 2, "lc3_recon_kernel": 8, "lc3_tns_kernel": 1, "lc3_decode_kernel": 4}  # (packer / parser: producer + consumer wave)


def load_pmc():
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            pj = json.load(f)
    except (OSError, ValueError):
        return None, "profiles/pmc_latest.json missing"
    if pj.get("kernel_source_sha256") != kernel_source_sha():
        return None, "profiles/pmc_latest.json was measured on other kernel sources (sha mismatch): counters withheld"
    return pj, None


# ---------------------------------------------------------------------------------------------------------------
# engines
# ---------------------------------------------------------------------------------------------------------------
class GpuEngine:
    """liblc3gpu.so through its C ABI; torch owns device memory and the HIP streams"""

    def __init__(self, args, pcm_host, S, T, mode, local_rank):
        import torch

        self.torch = torch
        torch.cuda.set_device(local_rank)
        assert torch.cuda.is_available(), "bench.py needs a HIP device (the engine has no CPU path)"
        pkg = importlib.import_module("lc3-codec_amd")
        self.pkg, self.S, self.T, self.mode = pkg, S, T, mode
        # pcm_host int16[S][R][nf], R a multiple of T: R / T "rotations" resident in HBM as int16[R / T][S][T][nf]; step k codes rotation
        # k mod (R / T) -- consecutive frames of every stream from step to step (the state is carried), a seam only where it wraps
        R = pcm_host.shape[1]
        assert R % T == 0
        self.n_rot = R // T
        full = torch.from_numpy(pcm_host).cuda()
        self.d_pcm_rot = full.reshape(S, self.n_rot, T, NF).permute(1, 0, 2, 3).contiguous()
        self.d_full = full if self.n_rot > 1 else None  # (kept for the other shapes of the batch, shape_bench: 1 GB of 288)
        self.d_bytes = torch.zeros((S, T, NBYTES), dtype=torch.uint8, device="cuda")
        self.d_out = torch.zeros((S, T, NF), dtype=torch.int16, device="cuda") if mode == "roundtrip" else None
        NP = max(1, args.hip_streams)
        assert S % NP == 0, "--streams must be a multiple of --hip-streams"
        self.NP, self.SP = NP, S // NP
        mk = lambda cls: [cls(self.SP, pkg.FrameDuration.TenMs, pkg.SamplingFrequency.Hz48000) for _ in range(NP)]
        self.encs = mk(pkg.Lc3Encoder)
        self.decs = mk(pkg.Lc3Decoder) if mode == "roundtrip" else []
        self.hs = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(NP - 1)]
        self.marks = []
        # the pipelined arrangement (roundtrip, one handle pair): encoder stream, decoder stream, two byte buffers
        self.arrangement = "single"
        self.k = 0
        self.pending = None
        if mode == "roundtrip" and NP == 1:
            # (LC3_BENCH_PRIO=enc|dec: that handle's stream gets the higher HIP stream priority -- a measurement aid)
            prio = os.environ.get("LC3_BENCH_PRIO", "")
            self.s_enc = torch.cuda.Stream(priority=-1) if prio == "enc" else torch.cuda.current_stream()
            self.s_dec = torch.cuda.Stream(priority=-1 if prio == "dec" else 0)
            self.bufs = [self.d_bytes, torch.zeros_like(self.d_bytes), torch.zeros_like(self.d_bytes)]
            self.enc_done = [torch.cuda.Event() for _ in range(3)]
            self.dec_done = [torch.cuda.Event() for _ in range(3)]
            # the staggered arrangement: the encoder records this event behind its back half (lc3gpu_encoder_stage_event), the decoder's
            # stream waits for it
            self.ev_back = torch.cuda.Event()
            self.ev_back.record(self.s_enc)  # (torch creates the HIP event at its first record)
            # the `split:..` arrangements: the batch as GROUPS of its streams, each group with an encoder handle and a decoder handle of its
            # own on one or two HIP streams of its own.  Handles, streams and events are created when an arrangement is first selected
            self.splits = {}
        elif mode == "encode" and NP == 1:
            self.bufs = [self.d_bytes, torch.zeros_like(self.d_bytes)]  # (the pipeline arrangement's encode-only form: two byte buffers)

    device = "cuda"
    carries_state = True

    def _pcm(self):
        """the PCM of the step about to be queued"""
        return self.d_pcm_rot[self.k % self.n_rot]

    def set_arrangement(self, name):
        """`pipeline`, `single`, `pipelined`, `staggered` or `split:..`; call between synchronised phases only"""
        assert name == "single" or (self.mode == "roundtrip" and self.NP == 1) or (name == "pipeline" and self.NP == 1)
        self.sync()
        self.arrangement, self.k = name, 0
        if name == "pipeline" and not hasattr(self, "pl"):
            # the library's own arrangement (lc3gpu_pipeline_*): handles, streams, priorities and events live behind the C ABI
            self.pl = self.pkg.Lc3Pipeline(self.S, self.pkg.FrameDuration.TenMs, self.pkg.SamplingFrequency.Hz48000)

        if self.mode == "roundtrip" and self.NP == 1:
            self.encs[0].stage_event(self.pkg.ENC_STAGE_BACK, self.ev_back if name == "staggered" else None)
        if name.startswith("split:") and name not in self.splits:
            # The chip has a handful of hardware queues (4 by default, GPU_MAX_HW_QUEUES) and the runtime deals HIP streams onto them; two
            # streams on one queue run one behind the other, and a stream waiting for an event holds up whatever shares its queue.  So
            # the arrangements draw their streams from ONE list per role (created as needed and shared by every split arrangement) ...
            torch, pkg = self.torch, self.pkg
            widths = [int(x) for x in name[6:].split("+")]
            G = len(widths)
            # ... and the ENCODER streams of the groups run at the higher HIP stream priority (round 5: with the encoder kernels
            # preferred when queues compete, the front halves -- the longest kernels of the critical encoder chains -- keep their pace and the
            # latency-bound kernels stretch instead: 55.8 against 54.5 M sustained on the round's final kernels; LC3_BENCH_PRIO=none|dec to
            # compare).  Streams of another priority live on hardware queues of their own.
            prio = os.environ.get("LC3_BENCH_PRIO", "enc")
            if not hasattr(self, "split_streams"):
                self.split_streams = {"enc": [self.s_enc] if prio != "enc" else [], "dec": [self.s_dec] if prio != "dec" else []}

            def stream_of(role, i):
                have = self.split_streams[role]
                while len(have) <= i:
                    have.append(torch.cuda.Stream(priority=-1 if prio == role else 0))
                return have[i]

            pool = []
            n_enc = n_dec = 0
            for w in widths:
                pool.append(stream_of("enc", n_enc))
                n_enc += 1
                if w == 2:
                    pool.append(stream_of("dec", n_dec))
                    n_dec += 1
            bounds = [((self.S // 4) * g // G) * 4 for g in range(G)] + [self.S]  # groups of whole workgroups (four streams each)
            groups, nxt = [], 0
            for g, w in enumerate(widths):
                lo, hi = bounds[g], bounds[g + 1]
                mk = lambda cls: cls(hi - lo, pkg.FrameDuration.TenMs, pkg.SamplingFrequency.Hz48000)
                groups.append({"lo": lo, "hi": hi, "enc": mk(pkg.Lc3Encoder), "dec": mk(pkg.Lc3Decoder), "s_enc": pool[nxt],
                               "s_dec": pool[nxt + w - 1], "enc_done": [torch.cuda.Event() for _ in range(2)],
                               "dec_done": [torch.cuda.Event() for _ in range(2)]})
                nxt += w
            self.splits[name] = groups

    def _handles(self):
        """the handles of the current arrangement"""
        if self.arrangement.startswith("split:"):
            gs = self.splits[self.arrangement]
            return [g["enc"] for g in gs], [g["dec"] for g in gs]
        if self.arrangement == "pipeline":
            return [g["enc"] for g in self.pl.groups], [g["dec"] for g in self.pl.groups]
        return self.encs, self.decs

    def _decode_pending(self, beside_packer):
        k = self.pending
        b = k % 3
        if beside_packer:
            self.s_dec.wait_event(self.ev_back)  # as recorded inside the encoder call of step k + 1, queued just before
        self.s_dec.wait_event(self.enc_done[b])
        self.decs[0].decode(self.bufs[b], self.d_out, NBYTES, self.T, stream=self.s_dec.cuda_stream)
        self.dec_done[b].record(self.s_dec)
        self.pending = None

    def step(self):
        d_pcm = self._pcm()
        if self.arrangement == "pipeline":
            if self.mode == "encode":  # (lc3gpu_pipeline_encode: the groups' encoder chains side by side)
                self.pl.encode(d_pcm, self.bufs[self.k & 1], NBYTES, self.T)
            else:
                self.pl.submit(d_pcm, self.bufs[self.k & 1], self.d_out, NBYTES, self.T)
            self.k += 1
            return
        if self.arrangement == "staggered":
            # like `pipelined`, one step further apart: step k's encoder call is queued, then the decoder of step k - 1 -- behind the point
            # where step k's encoder has only its packer left.  The parser (a lane per frame: a quarter of the chip's workgroup slots) then
            # runs beside the packer (likewise) and not beside the encoder's front half, which needs every slot; three byte buffers
            k, b = self.k, self.k % 3
            if k >= 3:
                self.s_enc.wait_event(self.dec_done[b])
            self.encs[0].encode(d_pcm, self.bufs[b], NBYTES, self.T, stream=self.s_enc.cuda_stream)
            self.enc_done[b].record(self.s_enc)
            if self.pending is not None:
                self._decode_pending(True)
            self.pending = k
            self.k += 1
            return
        if self.arrangement.startswith("split:"):
            # groups of the streams, each with its own handle pair: on two HIP streams `pipelined` by itself (two byte buffers, events), on
            # one HIP stream encode then decode in stream order.  The groups never join: each keeps its own encode -> decode -> encode
            # chain, so kernels of several calls can meet on the chip at any moment
            k, b = self.k, self.k & 1
            for g in self.splits[self.arrangement]:
                lo, hi, se, sd = g["lo"], g["hi"], g["s_enc"], g["s_dec"]
                if se is sd:
                    g["enc"].encode(d_pcm[lo:hi], self.bufs[b][lo:hi], NBYTES, self.T, stream=se.cuda_stream)
                    g["dec"].decode(self.bufs[b][lo:hi], self.d_out[lo:hi], NBYTES, self.T, stream=se.cuda_stream)
                    continue
                if k >= 2:
                    se.wait_event(g["dec_done"][b])
                g["enc"].encode(d_pcm[lo:hi], self.bufs[b][lo:hi], NBYTES, self.T, stream=se.cuda_stream)
                g["enc_done"][b].record(se)
                sd.wait_event(g["enc_done"][b])
                g["dec"].decode(self.bufs[b][lo:hi], self.d_out[lo:hi], NBYTES, self.T, stream=sd.cuda_stream)
                g["dec_done"][b].record(sd)
            self.k += 1
            return
        if self.arrangement == "pipelined":
            # step k: the encoder writes buffer k & 1 on its stream as soon as the decoder of step k - 2 has read it; the decoder follows
            # on ITS stream as soon as the bytes are there -- and meanwhile the encoder is already on step k + 1
            k, b = self.k, self.k & 1
            if k >= 2:
                self.s_enc.wait_event(self.dec_done[b])
            self.encs[0].encode(d_pcm, self.bufs[b], NBYTES, self.T, stream=self.s_enc.cuda_stream)
            self.enc_done[b].record(self.s_enc)
            self.s_dec.wait_event(self.enc_done[b])
            self.decs[0].decode(self.bufs[b], self.d_out, NBYTES, self.T, stream=self.s_dec.cuda_stream)
            self.dec_done[b].record(self.s_dec)
            self.k += 1
            return
        for p in range(self.NP):
            st = self.hs[p].cuda_stream
            lo, hi = p * self.SP, (p + 1) * self.SP
            self.encs[p].encode(d_pcm[lo:hi], self.d_bytes[lo:hi], NBYTES, self.T, stream=st)
            if self.decs:
                self.decs[p].decode(self.d_bytes[lo:hi], self.d_out[lo:hi], NBYTES, self.T, stream=st)
        self.k += 1

    def sync(self):
        if self.pending is not None:  # staggered: the last step's decoder call
            self._decode_pending(False)
        if hasattr(self, "pl"):
            self.pl.wait()
        self.torch.cuda.synchronize()

    def step_mark(self):
        """an event behind the step just queued (single-stream runs: on the launch stream; pipelined: behind its decoder): consecutive
        marks bracket one step"""
        if self.NP == 1:
            e = self.torch.cuda.Event(enable_timing=True)
            self._record(e)
            self.marks.append(e)

    def _record(self, e):
        """an event behind the step just queued"""
        if self.arrangement == "pipeline":
            # lc3gpu_pipeline_mark: on the pipeline's own last stream.  (NOT join + a stream of ours: a fifth stream that waits for events
            # shares a hardware queue with one of the pipeline's four and holds it up -- measured 52.4 against 58 M frames/s)
            e.record(self.torch.cuda.current_stream())  # torch creates the HIP event at its first record; re-recorded by the library below
            self.pl.mark(e)
        else:
            e.record(self._last_stream())

    def _last_stream(self):
        """the stream the step's last call was queued on"""
        if self.arrangement.startswith("split:"):
            return self.splits[self.arrangement][-1]["s_dec"]
        return self.s_dec if self.arrangement != "single" else self.hs[0]

    def step_times_ms(self):
        m, self.marks = self.marks, []
        return [m[i].elapsed_time(m[i + 1]) for i in range(len(m) - 1)]

    def reset(self):
        self.sync()
        for h in self.encs + self.decs + [g[k] for gs in getattr(self, "splits", {}).values() for g in gs for k in ("enc", "dec")]:
            h.reset()
        if hasattr(self, "pl"):
            self.pl.reset()
        self.k = 0

    def sample(self, k):
        """the k streams the parity gate looks at: the first k -- in a split arrangement the first k / G of every group"""
        import numpy as np

        if self.arrangement.startswith("split:"):
            gs = self.splits[self.arrangement]
            per = max(1, k // len(gs))
            return np.concatenate([np.arange(g["lo"], min(g["hi"], g["lo"] + per)) for g in gs])
        if self.arrangement == "pipeline":
            gs = self.pl.groups
            per = max(1, k // len(gs))
            return np.concatenate([np.arange(g["first"], g["first"] + min(g["n"], per)) for g in gs])
        return np.arange(k)

    def last_bytes_all(self):
        """the byte buffer of the most recent step (device tensor, every stream)"""
        if self.arrangement in ("pipelined", "pipeline") or self.arrangement.startswith("split:"):
            return self.bufs[(self.k - 1) & 1]
        if self.arrangement == "staggered":
            return self.bufs[(self.k - 1) % 3]
        return self.d_bytes

    def last_bytes(self, k):
        """the sampled streams' frame bytes of the most recent step"""
        return self.last_bytes_all()[self.torch.from_numpy(self.sample(k)).cuda()].cpu().numpy()

    def results(self, k):
        idx = self.torch.from_numpy(self.sample(k)).cuda()
        return (self.last_bytes(k), self.d_out[idx].cpu().numpy() if self.d_out is not None else None)

    def timing_start(self, every=1):
        encs, decs = self._handles()
        for h in encs + decs:
            h.timing(every)

    def timing_stop(self):
        ef = ev = eb = ep = en = dp = dr = dt = ds = dn = 0.0
        encs, decs = self._handles()
        for h in encs:
            a, v, b, p_, n = h.timing(False)
            ef, ev, eb, ep, en = ef + a, ev + v, eb + b, ep + p_, en + n
        for h in decs:
            a, r, t, b, n = h.timing_kernels(False)
            dp, dr, dt, ds, dn = dp + a, dr + r, dt + t, ds + b, dn + n
        en, dn = en / len(encs), dn / max(1, len(decs))  # launches per step = one per handle and kernel: scale to "per step"
        km = {"lc3_enc_front_kernel": ef / max(en, 1), "lc3_sns_vq_kernel": ev / max(en, 1),
              "lc3_enc_back_kernel": eb / max(en, 1), "lc3_pack_kernel": ep / max(en, 1)}
        if decs:
            km.update({"lc3_parse_kernel": dp / max(dn, 1), "lc3_recon_kernel": dr / max(dn, 1), "lc3_tns_kernel": dt / max(dn, 1), "lc3_decode_kernel": ds / max(dn, 1)})
        return km

    def timed_steps(self, steps, warmup, dist=None, kernel_events=0, marks=False):
        """`warmup` untimed steps, then exactly `steps` steps between two synchronisation points -> (seconds, per-kernel ms or None,
        sorted per-step ms)"""
        for _ in range(warmup):
            self.step()
        self.sync()
        if dist is not None:
            dist.barrier()
        self.sync()
        if kernel_events:
            self.timing_start(kernel_events)
        if marks:
            self.step_mark()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
            if marks:
                self.step_mark()
        self.sync()
        if dist is not None:
            dist.barrier()
        self.sync()
        elapsed = time.perf_counter() - t0
        kernel_ms = self.timing_stop() if kernel_events else None
        return elapsed, kernel_ms, sorted(self.step_times_ms())

    def sustained(self, seconds):
        """back-to-back steps of the current arrangement for at least `seconds`; beside them, on a stream of its own, a one-wave probe
        kernel (lc3gpu_clock_probe) reads the shader clock every few steps: delta(s_memtime) / delta(s_memrealtime) x 100 MHz"""
        torch, pkg = self.torch, self.pkg
        n_slots, every, depth = 256, 16, 48
        slots = torch.zeros((n_slots, 3), dtype=torch.int64, device="cuda")
        s_probe = torch.cuda.Stream()
        fences = []
        self.sync()
        t0 = time.perf_counter()
        steps = probes = 0
        while True:
            self.step()
            steps += 1
            if steps % every == 0:
                if probes < n_slots:
                    pkg.clock_probe(slots[probes], stream=s_probe.cuda_stream, spin=50000)
                    probes += 1
                e = torch.cuda.Event()
                self._record(e)
                fences.append(e)
                if len(fences) > depth // every:
                    fences.pop(0).synchronize()  # the host stays at most `depth` steps ahead of the chip
                if time.perf_counter() - t0 >= seconds:
                    break
        self.sync()
        el = time.perf_counter() - t0
        v = slots[:probes].cpu().numpy()
        mhz = sorted(100.0 * float(c) / float(r) for c, r, _ in v if r > 0)
        skip = len(mhz) // 4  # the first probes run while the clock is still ramping: the figures are of the later three quarters
        late = sorted(100.0 * float(c) / float(r) for c, r, _ in v[skip:] if r > 0)
        return {"seconds": el, "steps": steps, "value": self.S * self.T * steps / el, "unit": "frames/s", "ms_per_step": el / steps * 1e3,
                "arrangement": self.arrangement,
                "shader_clock_MHz": {"median": late[len(late) // 2] if late else None, "min": late[0] if late else None,
                                     "max": late[-1] if late else None, "probes": len(late),
                                     "how": "one-wave kernel on a HIP stream of its own beside the steps: delta(s_memtime) / delta(s_memrealtime) "
                                            "x 100 MHz over ~0.1 ms, every 16th step; first quarter of the probes dropped"}}


class EmuEngine:
    """tests only: the device headers under the CPU wave emulator (tests/emu), host arrays, fresh state every step"""

    device = "cpu"
    arrangement = "single"
    carries_state = False  # (every step starts from fresh state)

    def __init__(self, args, pcm_host, S, T, mode, local_rank):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import emu_lib

        self.E, self.pcm, self.mode = emu_lib, pcm_host, mode
        self.bytes = self.out = None

    def step(self):
        self.bytes = self.E.encode(self.pcm, NBYTES)
        if self.mode == "roundtrip":
            self.out = self.E.decode(self.bytes, NF)

    def sync(self):
        pass

    def set_arrangement(self, name):
        assert name == "single"

    def reset(self):
        pass

    def results(self, k):
        return self.bytes[:k], (self.out[:k] if self.out is not None else None)

    def timed_steps(self, steps, warmup, dist=None, kernel_events=0, marks=False):
        for _ in range(warmup):
            self.step()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        if dist is not None:
            dist.barrier()
        return time.perf_counter() - t0, None, []

    def sustained(self, seconds):
        return None


# ---------------------------------------------------------------------------------------------------------------
# SURVEY 8d, Config 2: the other shapes of the 65 536-frame batch, and the host-resident leg
# ---------------------------------------------------------------------------------------------------------------
def shape_bench(torch, pkg, d_full, pcm_host, S, T, steps, warmup, cold=False, arrangements=("pipeline", "single"), gate=True):
    """One shape S streams x T frames per step (S T = the headline's frames per step) on the PCM already resident in HBM: the headline's
    16 384 streams x R frames seen as S streams of L = R / ceil(S / 16 384) consecutive frames (a stream's R frames in that many
    pieces), step k on frames [T k, T k + T) mod L.  cold: every step starts from FRESH state (reset + encode + decode inside the timed step:
    "a new reference encoder / decoder per frame", /root/reference/src/encoder/lc3_encoder.rs:117-173); otherwise state is carried (a live
    server's tick when T = 1).  Each arrangement with its own parity gate (two steps from fresh state, the second against the oracle on
    256 streams) and the per-kernel HIP-event times of every tenth step.  -> dict"""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    n0, R = d_full.shape[0], d_full.shape[1]
    pieces = (S + n0 - 1) // n0
    L = R // pieces
    assert L >= T and S <= n0 * pieces
    n_rot = L // T
    view = d_full.reshape(n0 * pieces, L, NF)[:S, :n_rot * T]
    d_rot = view.reshape(S, n_rot, T, NF).permute(1, 0, 2, 3).contiguous()
    host = pcm_host.reshape(n0 * pieces, L, NF)
    d_bytes = [torch.zeros((S, T, NBYTES), dtype=torch.uint8, device="cuda") for _ in range(2)]
    d_out = torch.zeros((S, T, NF), dtype=torch.int16, device="cuda")
    FD, SF = pkg.FrameDuration.TenMs, pkg.SamplingFrequency.Hz48000
    out = {"streams": S, "frames_per_stream_per_step": T, "frames_per_step": S * T, "state": "fresh every step (reset inside the timed step)" if cold else "carried",
           "resident_frames_per_stream": n_rot * T}
    thr = granted_cpus()[0]
    for arr in arrangements:
        if arr == "pipeline":
            pl = pkg.Lc3Pipeline(S, FD, SF)
            encs, decs = [g["enc"] for g in pl.groups], [g["dec"] for g in pl.groups]
            idx = np.concatenate([np.arange(g["first"], g["first"] + min(g["n"], 256 // len(pl.groups))) for g in pl.groups])

            def step(k):
                if cold:
                    pl.reset()
                pl.submit(d_rot[k % n_rot], d_bytes[k & 1], d_out, NBYTES, T)

            wait, reset = pl.wait, pl.reset
        else:
            enc, dec = pkg.Lc3Encoder(S, FD, SF), pkg.Lc3Decoder(S, FD, SF)
            encs, decs = [enc], [dec]
            idx = np.arange(min(S, 256))
            st = torch.cuda.current_stream().cuda_stream

            def step(k):
                if cold:
                    enc.reset()
                    dec.reset()
                enc.encode(d_rot[k % n_rot], d_bytes[k & 1], NBYTES, T, stream=st)
                dec.decode(d_bytes[k & 1], d_out, NBYTES, T, stream=st)

            wait = torch.cuda.synchronize

            def reset():
                torch.cuda.synchronize()
                enc.reset()
                dec.reset()

        res = {}
        if gate:
            reset()
            step(0)
            step(1 % n_rot if n_rot > 1 else 1)
            wait()
            torch.cuda.synchronize()
            tidx = torch.from_numpy(idx).cuda()
            got_b, got_p = d_bytes[1][tidx].cpu().numpy(), d_out[tidx].cpu().numpy()
            second = host[idx, T:2 * T] if n_rot > 1 else host[idx, :T]
            if cold:
                ref_b = O.encode_batch(np.ascontiguousarray(second), NBYTES, FS, US, threads=thr)
                ref_p = O.decode_batch(ref_b, NF, FS, US, threads=thr)
            else:
                two = np.ascontiguousarray(host[idx, :2 * T] if n_rot > 1 else np.concatenate([host[idx, :T], host[idx, :T]], axis=1))
                rb = O.encode_batch(two, NBYTES, FS, US, threads=thr)
                ref_b, ref_p = rb[:, T:], O.decode_batch(rb, NF, FS, US, threads=thr)[:, T:]
            bad_b = int((got_b != ref_b).any(axis=2).sum())
            diff = np.abs(got_p.astype(np.int32) - ref_p.astype(np.int32))
            res["parity"] = {"frames_checked": int(len(idx) * T), "bitstream_exact": bad_b == 0, "pcm_max_abs_diff": int(diff.max())}
            res["parity_mismatches"] = bad_b + int((diff.max(axis=2) > 1).sum())
            reset()
        for k in range(warmup):
            step(k)
        wait()
        torch.cuda.synchronize()
        for h in encs + decs:
            h.timing(KERNEL_EVENTS_EVERY)
        t0 = time.perf_counter()
        for k in range(warmup, warmup + steps):
            step(k)
        wait()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        ef = ev = eb = ep = en = dp = ds = dn = 0.0
        for h in encs:
            a, v, b, p_, n = h.timing(False)
            ef, ev, eb, ep, en = ef + a, ev + v, eb + b, ep + p_, en + n
        for h in decs:
            a, _, _, b, n = h.timing_kernels(False)
            dp, ds, dn = dp + a, ds + b, dn + n
        en, dn = max(1.0, en / len(encs)), max(1.0, dn / len(decs))
        res.update({"value": S * T * steps / el, "unit": "frames/s", "ms_per_step": el / steps * 1e3,
                    "kernel_ms": {"lc3_enc_front_kernel": ef / en, "lc3_sns_vq_kernel": ev / en, "lc3_enc_back_kernel": eb / en, "lc3_pack_kernel": ep / en,
                                  "lc3_parse_kernel": dp / dn, "lc3_decode_kernel": ds / dn}})
        out[arr] = res
        if arr == "pipeline":
            pl.close()
        else:
            enc.close()
            dec.close()
    best = max((a for a in arrangements), key=lambda a: out[a]["value"])
    out["value"], out["arrangement"] = out[best]["value"], best
    return out


def host_resident_bench(torch, pkg, pcm_host, S, T, reps=3):
    """lc3gpu_encode_host / lc3gpu_decode_host on the headline batch with PINNED host buffers: PCM and bytes start and end in host memory,
    both PCIe copies inside the timed region (the caller of /root/reference/examples/encode.rs:73-116 holds its data there).  -> dict"""
    import numpy as np

    FD, SF = pkg.FrameDuration.TenMs, pkg.SamplingFrequency.Hz48000
    enc, dec = pkg.Lc3Encoder(S, FD, SF), pkg.Lc3Decoder(S, FD, SF)
    p_in = pkg.PinnedBuffer((S, T, NF), np.int16)
    p_b = pkg.PinnedBuffer((S, T, NBYTES), np.uint8)
    p_out = pkg.PinnedBuffer((S, T, NF), np.int16)
    p_in.array[...] = pcm_host[:, :T]
    te, td = [], []
    for _ in range(reps + 1):
        t0 = time.perf_counter()
        enc.encode_host(p_in.array, p_b.array, NBYTES, T)
        t1 = time.perf_counter()
        dec.decode_host(p_b.array, p_out.array, NBYTES, T)
        t2 = time.perf_counter()
        te.append(t1 - t0)
        td.append(t2 - t1)
    te, td = min(te[1:]), min(td[1:])
    for b in (p_in, p_b, p_out):
        b.close()
    return {"frames": S * T, "encode_host_frames_per_s": S * T / te, "decode_host_frames_per_s": S * T / td,
            "roundtrip_host_frames_per_s": S * T / (te + td), "encode_ms": te * 1e3, "decode_ms": td * 1e3,
            "pcie_GBs_encode": S * T * (2 * NF + NBYTES) / te / 1e9, "pcie_GBs_decode": S * T * (2 * NF + NBYTES) / td / 1e9,
            "what": "lc3gpu_encode_host then lc3gpu_decode_host (synchronous calls, pinned host buffers from lc3gpu_host_alloc): H2D copy, kernels and "
                    "D2H copy of channel ranges of 32 768 frames on three internal HIP streams (copy-in, kernels, copy-out side by side); best of %d repetitions; NOT `value` (inputs there are resident in HBM)" % reps}


# ---------------------------------------------------------------------------------------------------------------
def run_rank(args):
    import numpy as np

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    emu = args.engine == "emu"
    dist = None
    rccl_version = None
    json_out = sys.stdout
    # LC3_BENCH_RCCL=1: the RCCL path of an N-GPU run (library load, communicator, barrier, all_reduce on device tensors) with ONE rank on
    # the one GPU a box of this pool has -- entered before anything else touches the device, as the ranks of an 8-GPU run do
    rccl_single = world == 1 and not emu and os.environ.get("LC3_BENCH_RCCL") == "1"
    if world > 1 or rccl_single:
        import torch
        import torch.distributed as dist

        if emu:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        elif os.environ.get("LC3_BENCH_BACKEND") == "gloo":
            # test aid for boxes with fewer GPUs than ranks: the ranks share the visible devices and reduce over gloo (RCCL needs
            # one GPU per rank); everything else -- launcher, sharding, GPU engine -- is the production path
            local_rank = local_rank % max(1, torch.cuda.device_count())
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            # stdout carries ONE JSON line (rank 0).  RCCL writes its version banner (NCCL_DEBUG=VERSION, set on some boxes) and its
            # warnings to file descriptor 1 through C stdio, at times of its own choosing: from here on descriptor 1 IS stderr for
            # everybody, and the JSON line goes to a duplicate of the original stdout
            sys.stdout.flush()
            json_out = os.fdopen(os.dup(1), "w")
            os.dup2(2, 1)
            torch.cuda.set_device(local_rank)
            if rccl_single:
                dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                                        device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
            try:
                rccl_version = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:  # noqa: BLE001 -- informational only
                rccl_version = None

    if dist is not None and os.environ.get("LC3_BENCH_TEST_DIE_RANK") == str(rank):
        # test hook (tests/test_dist_gloo.py): this rank dies after the rendezvous; the launcher must stop the others, which are
        # about to wait for it in a collective
        dist.barrier()
        os._exit(17)
    synth = importlib.import_module("lc3-codec_amd.synth")
    D = importlib.import_module("lc3-codec_amd.dist")
    mode = args.mode
    if mode == "roundtrip":
        S = args.streams if args.streams is not None else (4 if emu else 16384)
        T = args.frames if args.frames is not None else (2 if emu else 4)
        first_stream, total_streams, scaling = rank * S, S * world, "weak"
    else:
        T = args.frames if args.frames is not None else (2 if emu else 16)
        total = args.frames_total if args.frames_total is not None else (16 if emu else 1048576)
        assert total % T == 0, "--frames-total must be a multiple of --frames"
        total_streams = total // T
        lo, hi = D.shard_range(total_streams, world, rank)
        S, first_stream, scaling = hi - lo, lo, "strong"
    frames_per_step = S * T

    # synthetic input: every stream of the headline batch is its own (16 384 distinct streams per rank; the lane-per-frame kernels take
    # as long as the longest frame of a wave, so tiled copies would under-sample that tail); larger batches tile 16 384.  Round trip: R = 64
    # consecutive frames of every stream are resident (1 GB of PCM per GPU) and step k codes frames [T k, T k + T) mod R, so that the timed
    # region walks through 0.64 s of every stream instead of looping over one 40 ms stretch (generated by all granted host cores: ~3 s)
    n_distinct = min(S, 16384)
    R = T
    if mode == "roundtrip" and not emu:
        R = args.resident_frames if args.resident_frames is not None else 64
        R = max(T, R // T * T)
    base = synth.make_pcm_parallel(n_distinct, R, NF, FS, first_stream=first_stream, workers=max(1, granted_cpus()[0] // max(1, world)))
    pcm_host = np.ascontiguousarray(np.tile(base, ((S + n_distinct - 1) // n_distinct, 1, 1))[:S])
    n_rot = R // T
    eng = (EmuEngine if emu else GpuEngine)(args, pcm_host, S, T, mode, local_rank)
    can_pipeline = (not emu) and mode == "roundtrip" and max(1, args.hip_streams) == 1
    main_arr = args.arrangement if can_pipeline else "single"
    if (not emu) and mode == "encode" and max(1, args.hip_streams) == 1 and args.arrangement == "pipeline":
        main_arr = "pipeline"  # (encode only: lc3gpu_pipeline_encode, +3 % over one stream; the other arrangements need a decoder)

    # parity gate, on the arrangement that is timed: two steps from fresh state queued back to back (the second one carries state, and
    # in the pipelined arrangement both byte buffers and all four events are in play), the SECOND step's bitstream and PCM against the
    # CPU oracle on a sample of the streams, every rank; the oracle runs on this rank's share of the host threads the job is granted
    def gate(arr):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O

        eng.set_arrangement(arr)
        eng.reset()
        eng.step()
        eng.step()
        eng.sync()
        # The sample does not depend on the number of ranks (every rank checks the same number of its own streams) and is sized so that the
        # oracle stays a small part of a rank's wall time however many ranks share the host: 256 streams x 2 T frames is ~70 ms of oracle
        # work per granted thread on 16 threads at N = 1, ~0.5 s on the 2 threads a rank of 8 gets
        k = min(n_distinct, 256)
        thr = max(1, granted_cpus()[0] // max(1, world))
        idx = eng.sample(k) if hasattr(eng, "sample") else np.arange(k)  # (split arrangements: streams of every group)
        k = len(idx)
        if n_rot >= 2:
            two = np.ascontiguousarray(pcm_host[idx, :2 * T])  # the two steps code consecutive frames
        else:
            two = np.ascontiguousarray(np.concatenate([pcm_host[idx], pcm_host[idx]], axis=1))  # every step codes the same T frames
        t_or = time.perf_counter()
        ref_b = O.encode_batch(two, NBYTES, FS, US, threads=thr)
        if os.environ.get("LC3_BENCH_TEST_CORRUPT_GATE") == "1":  # test hook: a gate that MUST fail (the run then has to exit non-zero)
            ref_b = ref_b.copy()
            ref_b[0, -1, 0] ^= 1
        got_b, got_p = eng.results(k)
        second = slice(T, 2 * T) if eng.carries_state else slice(0, T)
        bad_b = int((got_b != ref_b[:, second]).any(axis=2).sum())
        par = {"arrangement": arr, "frames_checked": int(k * T), "which": "second of two steps from fresh state (state carried)",
               "bitstream_exact": bad_b == 0, "oracle_threads": thr}
        mism = bad_b
        if mode == "roundtrip":
            ref_p = O.decode_batch(ref_b, NF, FS, US, threads=thr)
            diff = np.abs(got_p.astype(np.int32) - ref_p[:, second].astype(np.int32))
            par["pcm_max_abs_diff"] = int(diff.max())
            mism += int((diff.max(axis=2) > 1).sum())
        par["oracle_seconds"] = time.perf_counter() - t_or
        eng.reset()
        return par, mism

    parity, mismatches = None, 0
    if not args.no_parity:
        parity, mismatches = gate(main_arr)

    # timed region: exactly K steps; per-kernel durations from HIP events the C ABI records on the launch streams on every tenth
    # step (an event after every kernel of every step costs the streams ~0.03 ms per step), and one event after every step for the
    # per-step minimum / median
    eng.set_arrangement(main_arr)
    elapsed, kernel_ms, step_ms = eng.timed_steps(args.steps, args.warmup, dist=dist, kernel_events=KERNEL_EVENTS_EVERY, marks=True)

    # max time over ranks, counters summed over ranks (the only collective of the job)
    red_dev = "cpu" if (dist is not None and dist.get_backend() == "gloo") else eng.device
    elapsed, total_frames, total_mismatches, _ = D.reduce_report(dist, red_dev, elapsed, frames_per_step * args.steps,
                                                                  mismatches=mismatches, force=rccl_single)
    # the other arrangement (same K steps, its own parity gate) and the sustained leg of the main one: single rank only
    other, others, sustained = None, [], None
    if world == 1 and not emu:
        if can_pipeline and not args.no_overlap_probe:
            for arr2 in [a for a in ("single", "pipelined", "staggered") + tuple(args.also) if a != main_arr]:
                par2, mism2 = gate(arr2) if not args.no_parity else (None, 0)
                eng.set_arrangement(arr2)
                el2, km2, st2 = eng.timed_steps(args.steps, args.warmup, kernel_events=KERNEL_EVENTS_EVERY, marks=True)
                others.append({"arrangement": arr2, "value": frames_per_step * args.steps / el2, "unit": "frames/s", "ms_per_step": el2 / args.steps * 1e3,
                               "ms_per_step_median": st2[len(st2) // 2] if st2 else None, "kernel_ms": km2, "parity": par2, "parity_mismatches": mism2,
                               "hip_streams": arrangement_streams(arr2)})
                total_mismatches += mism2
            other = others[0]  # (the one-stream arrangement unless that is the timed one)
        if args.sustain_seconds > 0:
            eng.set_arrangement(main_arr)
            sustained = eng.sustained(args.sustain_seconds)
    other_modes, host_leg = None, None
    if world == 1 and not emu and mode == "roundtrip" and not args.no_other_modes and eng.d_full is not None and S * T == 65536 and S == 16384:
        # SURVEY 8d Config 2's own two modes (65 536 x 1 from fresh state, 4 096 x 16 streaming), the live-server tick (65 536 x 1, state
        # carried) and the T sweep between them, on the same resident PCM; the headline shape (16 384 x 4) is `value` above
        eng.sync()
        torch, pkg = eng.torch, eng.pkg
        other_modes = {"note": "the same 65 536 frames per step in other shapes (S streams x T frames), each with its own handles, parity gate (second of two "
                               "steps against the oracle, 256 streams) and per-kernel HIP-event times; `pipeline` = lc3gpu_pipeline_submit (two groups), "
                               "`single` = lc3gpu_encode then lc3gpu_decode on one stream; `value` = the better of the two"}
        other_modes["mode_a_cold_65536x1"] = shape_bench(torch, pkg, eng.d_full, pcm_host, 65536, 1, args.steps, args.warmup, cold=True)
        other_modes["tick_carried_65536x1"] = shape_bench(torch, pkg, eng.d_full, pcm_host, 65536, 1, args.steps, args.warmup)
        other_modes["mode_b_streaming_4096x16"] = shape_bench(torch, pkg, eng.d_full, pcm_host, 4096, 16, args.steps, args.warmup)
        sweep = []
        for T2 in (2, 4, 8, 32):
            r = shape_bench(torch, pkg, eng.d_full, pcm_host, 65536 // T2, T2, args.steps, args.warmup, arrangements=("pipeline",))
            sweep.append({"frames_per_stream_per_step": T2, "streams": 65536 // T2, "value": r["value"], "ms_per_step": r["pipeline"]["ms_per_step"],
                          "kernel_ms": r["pipeline"]["kernel_ms"], "parity_mismatches": r["pipeline"].get("parity_mismatches")})
            total_mismatches += r["pipeline"].get("parity_mismatches", 0)
        other_modes["t_sweep_pipeline"] = sweep
        for key in ("mode_a_cold_65536x1", "tick_carried_65536x1", "mode_b_streaming_4096x16"):
            for arr2 in ("pipeline", "single"):
                total_mismatches += other_modes[key][arr2].get("parity_mismatches", 0)
        host_leg = host_resident_bench(torch, pkg, pcm_host, S, T)

    if rank == 0:
        cpu = None
        if not args.no_cpu_baseline and world == 1 and not emu:
            cpu = cpu_baseline(mode)
        elif world > 1:
            # the contract times the CPU path on rank 0 at N = 1 only (N ranks share the host's cores with their own parity gates): the N = 1
            # line of the same command is where the figure is
            cpu = {"value": None, "unit": "frames/s", "cores": None, "kind": "port",
                   "sample": "not timed at N > 1: see `cpu_baseline` of `python bench.py --gpus 1` (same host, same oracle build)"}
        value = total_frames / elapsed
        roof = None
        if kernel_ms and max(kernel_ms.values()) > 0.0:
            # Algorithmic bytes per frame: the front half reads 2*nf of PCM, the packer writes nbytes, the
            # parser reads nbytes, the synthesis kernel writes 2*nf; the vector quantiser and the back half touch no
            # algorithmic bytes (their traffic is the planes between kernels).  The analysis of a frame is three kernels
            # (front half, quantiser, back half): ONE unit for the roofline, with the frame's PCM as its algorithmic bytes.
            # kernel_ms[k] = that kernel's launch durations summed over one step (one launch per kernel and step by default; with the
            # opt-in split of a call, LC3GPU_SPLIT=1, the sum over its part launches, as `rocprofv3 --kernel-trace --stats` totals them)
            alg_bytes = {"lc3_enc_front_kernel": 2 * NF, "lc3_sns_vq_kernel": 0, "lc3_enc_back_kernel": 0,
                         "lc3_pack_kernel": NBYTES, "lc3_parse_kernel": NBYTES, "lc3_recon_kernel": 0, "lc3_tns_kernel": 0, "lc3_decode_kernel": 2 * NF}
            groups = {"analysis (lc3_enc_front_kernel + lc3_sns_vq_kernel + lc3_enc_back_kernel)":
                      ["lc3_enc_front_kernel", "lc3_sns_vq_kernel", "lc3_enc_back_kernel"]}
            for k in kernel_ms:
                if k not in groups[next(iter(groups))]:
                    groups[k] = [k]
            group_ms = {g: sum(kernel_ms[k] for k in ks) for g, ks in groups.items()}
            dom = max(group_ms, key=group_ms.get)
            dom_ms, alg = group_ms[dom], sum(alg_bytes[k] for k in groups[dom])
            achieved = frames_per_step * alg / (dom_ms * 1e-3) / 1e9
            # counters of the committed PMC passes, scaled to this run's frames per launch; withheld (null) unless the file
            # was measured on exactly these kernel sources
            pj, pmc_note = load_pmc()
            traffic = traffic_step = valu_frac = valu_frac_vop2 = lane_frac = valu_insts = None
            if pj is not None:
                scale = frames_per_step / pj["frames_per_launch"]
                kk = pj["kernels"]
                live = [k for k in kernel_ms if k in kk and kernel_ms[k] > 0.002]  # kernels this run launched
                hbm = lambda k: (kk[k].get("fetch_size_kb_corrected", kk[k]["fetch_size_kb"]) + kk[k]["write_size_kb"]) * 1024.0 * scale
                traffic = sum(hbm(k) for k in groups[dom] if k in kk)
                traffic_step = sum(hbm(k) for k in live)
                valu_insts = {k: kk[k]["sq_insts_valu"] * scale for k in live}
                thread_cyc = sum(kk[k]["sq_thread_cycles_valu"] for k in live) * scale
                lane_frac = thread_cyc / (sum(valu_insts.values()) * 64.0)
                # Issue fractions of the step (profiles/r06_instr_cost_findings.txt).  Denominator: the SIMD-cycles of a step's
                # wall time at the maximum clock.  valu_frac: the hardware's own count of the time its vector ALUs were executing,
                # SQ_ACTIVE_INST_VALU (quad-cycles per wave, i.e. x 4) -- 4.0 cycles per instruction in every kernel of this codec, which is
                # what a wave can issue (one instruction per ~4.4 cycles whatever its class).  valu_frac_vop2: the same instructions at the
                # 2 cycles a SIMD needs for a full-rate instruction (v_add/mul/fma_f32, v_add_u32, ...) when two of its waves can issue.
                have = N_SIMD * (elapsed / args.steps) * CLOCK_MHZ * 1e6
                act = sum(kk[k].get("sq_active_inst_valu", kk[k]["sq_insts_valu"]) for k in live) * scale
                valu_frac = 4.0 * act / have
                valu_frac_vop2 = 2.0 * sum(valu_insts.values()) / have
                pmc_note = pj.get("source")
            # the dominant KERNEL alone on the chip (the one-stream arrangement's HIP-event times: nothing shares the chip with it there)
            alone = next((o["kernel_ms"] for o in others if o["arrangement"] == "single" and o.get("kernel_ms")), kernel_ms if main_arr == "single" else None)
            frac_alone = achieved_alone = None
            if alone and alone.get("lc3_enc_front_kernel", 0) > 0:
                achieved_alone = frames_per_step * 2 * NF / (alone["lc3_enc_front_kernel"] * 1e-3) / 1e9
                frac_alone = achieved_alone / HBM_PEAK_GBS
            roof = {
                "bound": "valu-issue", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "frac_dominant_kernel_alone": frac_alone, "achieved_dominant_kernel_alone": achieved_alone,
                "dominant_kernel_alone": "lc3_enc_front_kernel: 960 algorithmic bytes per frame / its launch duration in the one-stream arrangement",
                "traffic": traffic, "traffic_whole_step": traffic_step,
                "measured_copy_GBs": (load_ceiling() or {}).get("copy_GBs"),
                "algorithmic_bytes_per_frame": alg,
                "roundtrip_algorithmic_bytes_per_frame": ALG_BYTES_ENC + (ALG_BYTES_DEC if mode == "roundtrip" else 0),
                "roundtrip_achieved_GBs": value * (ALG_BYTES_ENC + (ALG_BYTES_DEC if mode == "roundtrip" else 0)) / 1e9 / world,
                "valu_frac": valu_frac, "valu_frac_vop2": valu_frac_vop2, "active_lane_frac": lane_frac, "valu_insts_per_step": valu_insts,
                "instruction_cost": "profiles/r06_instr_cost.json (tools/instr_cost.hip), reading in profiles/r06_instr_cost_findings.txt",
                "pmc": pmc_note,
                "note": "`bound` says what binds: vector-instruction issue, not HBM (SURVEY 8d honesty note: ~60 flop per algorithmic byte) -- achieved / peak / "
                        "frac are nevertheless the contract's HBM figures (algorithmic bytes over launch duration against 8 TB/s), `frac` with the kernels of "
                        "several calls sharing the chip as the timed arrangement has them, `frac_dominant_kernel_alone` for the front half alone on the chip. "
                        "achieved = algorithmic bytes of the dominant unit / the sum of its kernels' launch durations in a step (HIP events on the "
                        "launch streams; one launch per kernel and step); "
                        "traffic = FETCH_SIZE + WRITE_SIZE of the dominant unit per step in BYTES (FETCH_SIZE doubled for the kernels whose reads are "
                        "16-byte-per-lane coalesced, MI355X_MICROARCH.md), traffic_whole_step the same over every kernel of the step; measured_copy_GBs = "
                        "what a 16-byte-per-lane copy kernel reaches on this chip (read + write), beside the 8 TB/s vendor figure `frac` divides by; "
                        "valu_frac = 4 x sum over the step's kernels of SQ_ACTIVE_INST_VALU (the hardware's count of quad-cycles its waves spent executing vector "
                        "instructions: 1.00 per instruction in every kernel here) / (1024 SIMDs x the step's wall time at 2.4 GHz); valu_frac_vop2 = 2 x SQ_INSTS_VALU "
                        "over the same: the step's vector instructions at the 2-cycle rate a SIMD reaches for full-rate instructions with two waves issuing "
                        "(profiles/r06_instr_cost.json); active_lane_frac = SQ_THREAD_CYCLES_VALU / "
                        "(64 x SQ_INSTS_VALU); counters from the committed rocprofv3 PMC passes (profiles/pmc_latest.json), null when they "
                        "were taken on other kernel sources",
            }
        if mode == "roundtrip":
            workload = (f"{frames_per_step}-frame batch mono 48 kHz / 10 ms / 150-byte frames, encode+decode (BASELINE configs[1]); "
                        f"{n_distinct} distinct synthetic streams" + (f" tiled to {S}" if n_distinct < S else ""))
            metric = "LC3 frames/sec (encode+decode) @48kHz/10ms"
        else:
            workload = (f"{total_streams * T}-frame batch mono 48 kHz / 10 ms / 150-byte frames, encode only, streams sharded over "
                        f"{world} GPU(s) (BASELINE configs[2]); {n_distinct} distinct synthetic streams per rank" + (f" tiled to {S}" if n_distinct < S else ""))
            metric = "LC3 frames/sec (encode) @48kHz/10ms"
        hip_streams = max(1, args.hip_streams) if main_arr == "single" else arrangement_streams(main_arr)
        line = {
            "metric": metric, "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_min": step_ms[0] if step_ms else None, "ms_per_step_median": step_ms[len(step_ms) // 2] if step_ms else None,
            "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": workload, "mode": mode, "streams_per_gpu": S, "frames_per_stream_per_step": T,
                "frames_per_step_per_gpu": frames_per_step, "nbytes": NBYTES, "state": "carried across steps (streaming)",
                "resident_frames_per_stream": R, "resident_pcm_bytes_per_gpu": int(S) * int(R) * NF * 2,
                "rotation": f"step k codes frames [{T} k, {T} k + {T}) mod {R} of every stream" if n_rot > 1 else "every step codes the same frames",
                "parallelism": f"streams sharded over {world} GPU(s), no data-path collective; "
                               + (f"torch.distributed world size {dist.get_world_size()} ({dist.get_backend()}"
                                  + (f", RCCL {rccl_version}" if rccl_version else "") + ")" if dist is not None else "single process"),
                "hip_streams": hip_streams, "arrangement": main_arr,
                "arrangement_note": ("`pipeline`: lc3gpu_pipeline_submit -- the library's own object for the caller loop: two groups of the streams, each with an "
                                     "encoder handle on a high-priority HIP stream and a decoder handle on a stream of its own, two byte buffers alternating "
                                     "(what `quad` = split:2+2 builds by hand in this script); "
                                     "`single`: lc3gpu_encode then lc3gpu_decode of the same batch on ONE caller stream, every call behind the one before "
                                     "it; `pipelined`: the encoder handle on one caller stream, the decoder handle on another, two byte buffers and "
                                     "events (INTEGRATION.md, recommended caller pattern): the decoder works on step k while the encoder runs step k + 1; "
                                     "`staggered`: the same with three byte buffers and the decoder call of step k queued behind the point where the "
                                     "encoder call of step k + 1 has only its packer left (lc3gpu_encoder_stage_event): parser beside packer; "
                                     "`split:a+b..` (`quad` = 2+2, `tri` = 2+1, `duo` = 1+1): the streams in equal groups, each group with its own encoder handle and "
                                     "decoder handle on 2 caller streams (`pipelined` by itself) or 1 (encode then decode) -- the groups never join.  "
                                     "`value` is the arrangement named here; the other one is timed in the same run (`other_arrangement`)"),
                "engine": args.engine,
            },
            "kernel_ms": kernel_ms,
            "kernel_ms_from": f"HIP events around every kernel on every {KERNEL_EVENTS_EVERY}th step of the timed region, on the streams the kernels are "
                              "launched on; per step, summed over a kernel's launches",
            "roofline": roof, "cpu_baseline": cpu, "other_arrangement": other, "other_arrangements": others,
            "other_modes": other_modes, "host_resident": host_leg,
            "value_single_stream": (value if main_arr == "single" else (other["value"] if other and other["arrangement"] == "single" else None)),
            "sustained": sustained,
            "parity": parity, "parity_mismatches_all_ranks": total_mismatches,
        }
        print(json.dumps(line), file=json_out, flush=True)
    if dist is not None:
        dist.destroy_process_group()
    # a run whose parity gate found a difference is a FAILED run: the line above says so (`parity_mismatches_all_ranks`), and so does
    # the exit code of every rank (the count is reduced over the ranks), so that a driver looking at `rc` never books its `value`
    return 3 if total_mismatches > 0 else 0


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the parent only launches: no torch import, no HIP call in this process
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    sys.exit(run_rank(args))


if __name__ == "__main__":
    main()
