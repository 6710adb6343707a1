"""Deterministic synthetic PCM for parity tests and bench.py.

Host-side, numpy only: the same int16 arrays feed the CPU oracle and the GPU engine.
Per stream: a harmonic tone (f0 ~ logU[80, 400] Hz, partials 1, 1/2, 1/3 with random phases;
exercises pitch search / LTPF), level ~U[-18, -3] dBFS, additive uniform white noise at
U[-45, -20] dBFS, and with probability 1/16 per frame a 2 ms exponentially decaying click at
+6 dB (attack detector / TNS).  5 % of the streams are all-zero (zero-frame paths) and 5 % are
full-scale white noise (clipping, high global gain).  Computed in float64, rounded to nearest,
clipped to int16.  Layout int16[S][T][nf] (stream-major, planar)."""
import numpy as np

SEED = 0x4C43335F  # "LC3_"


def make_pcm(n_streams, n_frames, nf, fs_hz, seed=SEED, first_stream=0):
    """Streams are generated independently from (seed, stream index) so that shards of a larger
    batch (multi-GPU ranks) see exactly the streams a single-process run would."""
    out = np.zeros((n_streams, n_frames, nf), np.int16)
    n = n_frames * nf
    t = np.arange(n, dtype=np.float64) / float(fs_hz)
    for i in range(n_streams):
        sid = first_stream + i
        rng = np.random.default_rng([seed, sid])
        kind = rng.random()
        if kind < 0.05:
            continue  # silent stream
        if kind < 0.10:
            x = rng.uniform(-1.0, 1.0, n) * 40000.0  # clips
        else:
            f0 = float(np.exp(rng.uniform(np.log(80.0), np.log(400.0))))
            level = 10.0 ** (rng.uniform(-18.0, -3.0) / 20.0) * 32767.0
            noise = 10.0 ** (rng.uniform(-45.0, -20.0) / 20.0) * 32767.0
            ph = rng.uniform(0.0, 2.0 * np.pi, 3)
            x = np.zeros(n)
            for h, amp in enumerate((1.0, 0.5, 1.0 / 3.0)):
                x += amp * np.sin(2.0 * np.pi * f0 * (h + 1) * t + ph[h])
            x *= level / 1.6
            x += rng.uniform(-1.0, 1.0, n) * noise
            clicks = rng.random(n_frames) < (1.0 / 16.0)
            tau = 0.002 * fs_hz / 4.0
            for fr in np.flatnonzero(clicks):
                start = fr * nf + int(rng.integers(0, nf))
                ln = min(n - start, int(0.002 * fs_hz) * 3)
                env = np.exp(-np.arange(ln) / tau)
                x[start:start + ln] += 2.0 * level * env * rng.choice([-1.0, 1.0])
        out[i] = np.clip(np.rint(x), -32768, 32767).astype(np.int16).reshape(n_frames, nf)
    return out


def make_pcm_parallel(n_streams, n_frames, nf, fs_hz, seed=SEED, first_stream=0, workers=None):
    """make_pcm over `workers` child processes (streams are generated independently from (seed, stream index), so the result is the same
    array).  The children are fresh interpreters running THIS FILE as a script (numpy only) -- nothing is forked: the caller may hold
    a GPU context -- and hand their chunk back through a temporary .npy file.  Falls back to the serial generator if a child fails."""
    import os
    import subprocess
    import sys
    import tempfile

    if workers is None:
        try:
            workers = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            workers = os.cpu_count() or 1
    workers = max(1, min(int(workers), n_streams // 256))
    if workers <= 1:
        return make_pcm(n_streams, n_frames, nf, fs_hz, seed=seed, first_stream=first_stream)
    step = (n_streams + workers - 1) // workers
    out = np.empty((n_streams, n_frames, nf), np.int16)
    with tempfile.TemporaryDirectory(prefix="lc3synth") as d:
        procs = []
        for k, a in enumerate(range(0, n_streams, step)):
            n = min(step, n_streams - a)
            path = os.path.join(d, "c%d.npy" % k)
            procs.append((a, n, path, subprocess.Popen([sys.executable, os.path.abspath(__file__), str(n), str(n_frames), str(nf), str(fs_hz),
                                                        str(seed), str(first_stream + a), path])))
        ok = True
        for a, n, path, pr in procs:
            ok = (pr.wait() == 0) and ok
            if ok:
                out[a:a + n] = np.load(path)
    if not ok:
        return make_pcm(n_streams, n_frames, nf, fs_hz, seed=seed, first_stream=first_stream)
    return out


def make_ltpf_pcm(nf, fs_hz, n_frames=14):
    """Three streams that walk the decoder's long-term post-filter through all five of its frame-to-frame transition
    cases (decoder/long_term_post_filter.rs:142-160): a tone whose pitch glides (filter stays on while its lag changes),
    a tone that is interrupted by noise (filter switches off and on again) and a steady tone (unchanged filter)."""
    n = n_frames * nf
    t = np.arange(n, dtype=np.float64) / float(fs_hz)
    x = np.zeros((3, n))
    f = 180.0 + 60.0 * np.clip((t - 0.04) / 0.06, 0.0, 1.0)
    ph = 2.0 * np.pi * np.cumsum(f) / float(fs_hz)
    x[0] = 9000.0 * np.sin(ph) + 3000.0 * np.sin(2.0 * ph)
    tone = 9000.0 * np.sin(2.0 * np.pi * 220.0 * t) + 4000.0 * np.sin(2.0 * np.pi * 440.0 * t + 1.0)
    gate = ((t < 0.05) | (t > 0.09)).astype(np.float64)
    rng = np.random.default_rng([SEED, 5])
    x[1] = tone * gate + rng.uniform(-1.0, 1.0, n) * 3000.0 * (1.0 - gate)
    x[2] = 12000.0 * np.sin(2.0 * np.pi * 150.0 * t) + 5000.0 * np.sin(2.0 * np.pi * 300.0 * t + 0.3)
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16).reshape(3, n_frames, nf)


def make_bandlimited_pcm(n_streams, n_frames, nf, fs_hz, cutoff_hz, seed=SEED):
    """Streams whose content stops below `cutoff_hz` (tones with a random roll-off towards the cut-off, nothing above it):
    what an up-sampled narrower-band signal looks like to the bandwidth detector (encoder/bandwidth_detector.rs:64-127), so
    that its cut-off stage, and the TNS band layouts of the lower bandwidth indices, are exercised at a high sampling rate."""
    n = n_frames * nf
    t = np.arange(n, dtype=np.float64) / float(fs_hz)
    out = np.zeros((n_streams, n_frames, nf), np.int16)
    for i in range(n_streams):
        rng = np.random.default_rng([seed, 7919, i])
        k = int(rng.integers(12, 40))
        f = np.sort(rng.uniform(100.0, cutoff_hz, k))
        edge = rng.uniform(0.03, 0.6)                       # width of the roll-off below the cut-off, as a fraction of it
        slope = rng.uniform(5.0, 45.0)                      # dB lost across the roll-off
        rel = np.clip((f - cutoff_hz * (1.0 - edge)) / (cutoff_hz * edge), 0.0, 1.0)
        amp = 10.0 ** (-slope * rel / 20.0) * rng.uniform(0.3, 1.0, k)
        x = np.zeros(n)
        for fk, ak in zip(f, amp):
            x += ak * np.sin(2.0 * np.pi * fk * t + rng.uniform(0.0, 2.0 * np.pi))
        x /= max(1e-9, np.abs(x).max())
        # band-limited clicks (all partials in phase at t0, a short Gaussian envelope): temporal structure for the TNS stage
        grid = np.arange(200.0, cutoff_hz * 0.97, 150.0)
        for t0 in rng.uniform(0.0, n / float(fs_hz), max(1, n_frames // 2)):
            env = np.exp(-0.5 * ((t - t0) / 0.0015) ** 2)
            sel = env > 1e-4
            click = np.zeros(int(sel.sum()))
            for fk in grid:
                click += np.cos(2.0 * np.pi * fk * (t[sel] - t0))
            x[sel] += 2.0 * env[sel] * click / len(grid)
        x *= 10.0 ** (rng.uniform(-20.0, -6.0) / 20.0) * 32767.0 / max(1e-9, np.abs(x).max())
        out[i] = np.clip(np.rint(x), -32768, 32767).astype(np.int16).reshape(n_frames, nf)
    return out


if __name__ == "__main__":  # a worker of make_pcm_parallel: n_streams n_frames nf fs_hz seed first_stream out.npy
    import sys

    _a = sys.argv[1:]
    np.save(_a[6], make_pcm(int(_a[0]), int(_a[1]), int(_a[2]), int(_a[3]), seed=int(_a[4]), first_stream=int(_a[5])))
