"""Seeded random IAMF streams for the decoder facade: one or two elements of random kinds, a random output layout, bit depth,
sample format, frame size, gains, ramps, trims, rates.  TEST INFRASTRUCTURE.

oracle/gen_golden_fuzz.py decodes every stream with the REAL reference (oracle/_ref/libiamf_ref.so) and stores, per seed, the
per-call return values and a SHA-256 of the PCM in tests/golden/fuzz.json — a few dozen bytes per stream, so the set can be
wide; tests/test_gpu_fuzz_facade.py decodes the same streams through libiamf_hip.so and compares both.  (The 80 + 20 + 25
streams whose PCM is stored sample by sample are in e2e.npz / lfe.npz / tv.npz: a hash says THAT something differs, they
say where.)"""
import hashlib

import numpy as np

import e2e_cases as E

KINDS = ["stereo", "l51", "l512", "l514", "l71", "l712", "l714", "l312", "mono", "l714dmx", "scalable", "toa_projection",
         "zoa", "foa", "soa", "toa"]
MODES = [0, 1, 2, 4, 5, 6]
RATES = [(44100, 48000), (48000, 44100), (32000, 48000), (16000, 48000), (96000, 48000)]
N_SEEDS = 240
# the reference's two other builds: -DDISABLE_LFE_HOA=0 (oracle/_ref_lfe; the facade after iamf_hip_decoder_set_hoa_lfe) and
# -DSAMSUNG_TV (oracle/_ref_tv; iamf_hip_decoder_set_variant: other layout -> layout tables, a 12-channel PCM stride)
# "wide": the default build again with the dimensions the first generator held fixed — scalable layer stacks, any layout
# with demixing info, the demixing defaults, big-endian samples, more stream / output rates (with and without conversion),
# frames trimmed away completely, frame sizes that are not multiples of 4
VARIANTS = dict(default=(0, N_SEEDS), lfe=(100000, 120), tv=(200000, 120), wide=(300000, 240))
STACKS = [[1, 3, 7], [0, 1, 2, 5], [1, 8], [2, 4], [3, 4], [8, 3, 6], [1, 2], [1, 2, 3, 4], [2, 3], [1, 5], [2, 5, 6, 7], [0, 1],
          [1, 8, 3, 7], [2, 7], [0, 1, 8, 3, 4], [1, 2, 5, 6], [5, 7], [8, 6], [1, 7], [0, 2]]
WIDE_RATES = [(44100, 44100), (32000, 32000), (16000, 16000), (48000, 16000), (48000, 32000), (48000, 24000), (48000, 8000),
              (48000, 12000), (44100, 32000), (96000, 44100), (16000, 44100), (32000, 16000)] + RATES
SCENE = ["zoa", "foa", "soa", "toa", "toa_projection"]


def case(seed, variant="default"):
    """the e2e_cases-style description of stream `seed` of a variant's set"""
    rng = np.random.default_rng(900000 + VARIANTS[variant][0] + seed)
    pick = lambda xs: xs[int(rng.integers(0, len(xs)))]
    kinds = KINDS + 3 * SCENE if variant == "lfe" else KINDS   # the LFE generator works on scene-based elements
    two = rng.random() < 0.6
    pair = (pick(kinds), pick(kinds)) if two else (pick(kinds),)
    lay = int(rng.integers(0, 14))
    layout = ("binaural",) if lay == 13 else ("ss", lay)
    fs = pick([1024, 1024, 1024, 1024, 2048, 512, 960, 256, 240, 128])
    if "scalable" in pair or "toa_projection" in pair:
        fs = pick([1024, 1024, 2048, 512, 960])   # (the demixer's cross-fade windows are as long as a frame)
    frames = int(rng.integers(4, 9)) if fs >= 512 else int(rng.integers(12, 30))
    c = dict(layout=layout, bit_depth=pick([16, 16, 24, 32]), frames=frames, fs=fs, seed=910000 + 7 * seed,
             sample_size=pick([16, 16, 24, 32]), pair=pair,
             element_gain_q78=int(rng.integers(-1500, 300)), element2_gain_q78=int(rng.integers(-1500, 300)),
             output_gain_q78=int(rng.integers(-600, 300)),
             dmx_modes=[pick(MODES) for _ in range(32)], dmx_modes2=[pick(MODES) for _ in range(32)],
             scalable_modes1=[pick(MODES) for _ in range(32)], scalable_modes2=[pick(MODES) for _ in range(32)],
             recon_salt2=int(rng.integers(1, 50)))
    if rng.random() < 0.25:
        c["pair_ramps"] = True
    trims = {}
    if rng.random() < 0.3:
        trims[0] = (int(rng.integers(1, fs)), 0)
    if rng.random() < 0.3:
        trims[frames - 1] = (0, int(rng.integers(1, fs)))
    if trims:
        c["trims"] = trims
    if rng.random() < 0.2:
        c["rate"], c["out_rate"] = pick(RATES)
        # Sound System H from a scene-based element: render_H2M never writes slot 23 (h2m_rdr.c:1103-1150, the LFE2 slot is
        # not reserved for H), so the reference hands out what its frame buffer held before — silence without a resampler
        # (the buffers only ever rotate zeros into that slot), the previous frame's INTERLEAVED resampler output with one
        # (iamf_resample uses the frame buffer as scratch, IAMF_decoder.c:3235-3244).  Stale memory, not a result: this
        # library writes silence, and the combination is kept out of the comparison.
        if layout == ("ss", 7) and any(k in ("zoa", "foa", "soa", "toa", "toa_projection") for k in pair):
            del c["rate"], c["out_rate"]
    if rng.random() < 0.2:
        c["loudness"] = float(pick([-16.0, -24.0, -31.0]))
        c["mix_loudness_q78"] = int(rng.integers(-30, -10)) * 256
    if rng.random() < 0.15:
        c["limiter"] = False
    elif rng.random() < 0.3:
        c["threshold"] = float(pick([-3.0, -6.0, -0.5]))
    if variant == "wide":
        kinds2 = KINDS + ["scalable"] * 4 + ["dmx:%d" % l for l in (2, 3, 4, 5, 6, 7, 8, 1)]
        pair = tuple(pick(kinds2) for _ in pair)
        c["pair"] = pair
        for k in ("1", "2"):
            st = pick(STACKS)
            c["scalable_layers" + k] = st
            c["scalable_gains" + k] = {int(li): (int(rng.integers(1, 64)), int(rng.integers(-1200, 600)))
                                       for li in range(len(st)) if rng.random() < 0.4}
            c["dmx_default" + k] = (pick(MODES), int(rng.integers(0, 11)))
        if rng.random() < 0.25:
            c["big_endian"] = True
        c.pop("rate", None), c.pop("out_rate", None)
        if rng.random() < 0.35:
            c["rate"], c["out_rate"] = pick(WIDE_RATES)
            if layout == ("ss", 7) and any(k in SCENE for k in pair):   # (slot 23 of H behind the resampler: see above)
                c["out_rate"] = c["rate"] if c["rate"] in (16000, 32000, 44100, 48000) else 48000
                c["rate"] = c["out_rate"]
        if not any(k in ("scalable", "toa_projection") for k in pair) and rng.random() < 0.25:
            c["fs"] = fs = pick([1000, 441, 250, 1023, 77, 2000])
            c["frames"] = frames = int(rng.integers(4, 9)) if fs >= 441 else int(rng.integers(12, 30))
            c.pop("trims", None)
            trims = {}
        if rng.random() < 0.2:   # a frame trimmed away completely, or all but one sample
            f0 = int(rng.integers(0, frames))
            a = int(rng.integers(0, fs + 1))
            tr = dict(c.get("trims", {}))
            tr[f0] = (a, fs - a) if rng.random() < 0.6 else (a, max(0, fs - a - 1))
            c["trims"] = tr
    return c


def build(seed, variant="default"):
    name = "fuzz_%s_%d" % (variant, seed)
    E.CASES[name] = case(seed, variant)
    try:
        return E.build(name)[0], E.CASES[name]
    finally:
        del E.CASES[name]


def decode_kwargs(c, variant="default"):
    kw = dict(bit_depth=c["bit_depth"], out_rate=c.get("out_rate", 0), loudness=c.get("loudness", 0.0),
              limiter=c.get("limiter", True), threshold=c.get("threshold", -1.0))
    if variant == "tv":
        kw["pcm_channels"] = 12   # IAMF_decoder.c:3492-3495
    return kw


def digest(pcm):
    return hashlib.sha256(np.ascontiguousarray(pcm).tobytes()).hexdigest()
