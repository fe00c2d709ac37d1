"""Cases of the HOA LFE generator (reference h2m_rdr.c:1151-1239, enabled by the
reference's own build switch -DDISABLE_LFE_HOA=0): stage level (render_H2M with a filter carried over
several calls) and end to end (synthetic LPCM .iamf streams through IAMF_decoder_*).  Shared by
oracle/gen_golden_lfe.py (which asks the real reference) and the tests (oracle, HIP path)."""
import numpy as np

import iamf_writer as W
import synth

SS = dict(A=0x020, B=0x050, C=0x250, D=0x450, E=0x451, F=0x370, G=0x490, H=0x9A3, I=0x070,
          J=0x470, L312=0x312, L712=0x712)
SS_ENUM = dict(A=0, B=1, C=2, D=3, E=4, F=5, G=6, H=7, I=8, J=9, L712=10, L312=11)   # IAMF_SoundSystem

# name: (ambisonics order, output rendering id, sample rate of lfefilter_init, call sizes, seed)
STAGE = {
    "toa_B": (3, SS["B"], 48000, [320, 64, 1024, 7], 101),
    "toa_J": (3, SS["J"], 48000, [1024, 1024], 102),
    "toa_F_two_lfe": (3, SS["F"], 48000, [512, 33, 479], 103),
    "toa_H": (3, SS["H"], 48000, [256, 256], 104),
    "toa_G": (3, SS["G"], 48000, [300], 105),
    "toa_E": (3, SS["E"], 48000, [300], 106),
    "toa_D": (3, SS["D"], 48000, [128, 128], 107),
    "soa_312": (2, SS["L312"], 48000, [400], 108),
    "foa_B": (1, SS["B"], 48000, [640, 640], 109),
    "zoa_C": (0, SS["C"], 48000, [1000], 110),
    "toa_B_44k1": (3, SS["B"], 44100, [441, 1024], 111),
    "toa_I_16k": (3, SS["I"], 16000, [960], 112),
    "toa_A_no_lfe_slot": (3, SS["A"], 48000, [200], 113),
}


def programme(seed, channels, ns, rate=48000, silence_from=None):
    """Gaussian noise on every channel plus low-frequency content on W (a 50 Hz and a 90 Hz tone, a slow
    sweep) so that the 120 Hz low-pass has something to pass; optional digital silence from a sample on
    (the filter then decays through the denormal range)."""
    x = synth.gaussian(seed, channels, ns, 0.12)
    t = np.arange(ns, dtype=np.float64) / rate
    lf = 0.35 * np.sin(2 * np.pi * 50.0 * t) + 0.2 * np.sin(2 * np.pi * 90.0 * t + 0.3) \
        + 0.15 * np.sin(2 * np.pi * (20.0 + 200.0 * t) * t)
    x[0] = (x[0] + lf.astype(np.float32)).astype(np.float32)
    if silence_from is not None:
        x[:, silence_from:] = 0.0
    return np.clip(x, -1.0, 1.0 - 2.0 ** -15).astype(np.float32)


def stage_input(name):
    order, _, rate, sizes, seed = STAGE[name]
    return programme(seed, (order + 1) ** 2, sum(sizes), rate)


# end to end: name -> dict(order, layout (IAMF_SoundSystem name), bit_depth, frames, fs, seed, rate, silence_from)
E2E = {
    "toa_B_s16": dict(order=3, ss="B", bit_depth=16, frames=6, fs=1024, seed=201),
    "toa_J_s24": dict(order=3, ss="J", bit_depth=24, frames=4, fs=1024, seed=202),
    "foa_F_s16": dict(order=1, ss="F", bit_depth=16, frames=5, fs=960, seed=203),
    "toa_H_s16": dict(order=3, ss="H", bit_depth=16, frames=3, fs=1024, seed=204),
    "soa_C_441_to_48k": dict(order=2, ss="C", bit_depth=16, frames=5, fs=1024, seed=205, rate=44100),
    "toa_B_decay_s32": dict(order=3, ss="B", bit_depth=32, frames=14, fs=1024, seed=206, silence_from=2500),
    "toa_A_s16": dict(order=3, ss="A", bit_depth=16, frames=3, fs=1024, seed=207),   # no LFE slot: unchanged path
    # projection-mode ambisonics: W is channel 0 AFTER the de-mapping (IAMF_core_decoder.c:116-130)
    "toa_projection_D_s16": dict(order=3, ss="D", bit_depth=16, frames=4, fs=1024, seed=208, projection=True),
    # round 3: frames shorter than the limiter's delay and longer than 1024, the 14-channel system, the LFE pair of E
    "toa_B_fs256": dict(order=3, ss="B", bit_depth=16, frames=20, fs=256, seed=209),
    "soa_J_fs2048": dict(order=2, ss="J", bit_depth=16, frames=3, fs=2048, seed=210),
    "toa_G_s16": dict(order=3, ss="G", bit_depth=16, frames=4, fs=1024, seed=211),
    "foa_E_s24": dict(order=1, ss="E", bit_depth=24, frames=4, fs=1024, seed=212),
    "toa_D_s16": dict(order=3, ss="D", bit_depth=16, frames=5, fs=1024, seed=213),
    # trimmed frames: the reference renders a frame and trims the result (IAMF_decoder.c:3424-3430), so the generator's filter
    # runs over the samples that are cut (found by tests/test_gpu_fuzz_facade.py; the batch takes them as
    # iamf_hip_render_args::lfe_pre_samples / lfe_post_samples)
    "toa_B_trim_s16": dict(order=3, ss="B", bit_depth=16, frames=6, fs=1024, seed=221, trims={0: (100, 0), 2: (7, 0), 5: (0, 300)}),
    "foa_F_trim_s24": dict(order=1, ss="F", bit_depth=24, frames=5, fs=960, seed=222, trims={0: (959, 0), 4: (0, 1)}),
    # round 4: presentations of TWO elements.  A scene-based element beside a channel-based one (plain or scalable), in
    # either position; and two scene-based elements: the reference keeps ONE filter per output layout
    # (IAMF_decoder.c:2629-2632: plfe = &stream->final_layout->sp.lfe_f), so the two W channels run through the same
    # two histories in turn, frame by frame, in presentation order
    "toa_plus_stereo_B_s16": dict(order=3, ss="B", bit_depth=16, frames=5, fs=1024, seed=214, second=("stereo",), gains=(-300, -500)),
    "stereo_plus_toa_J_s16": dict(order=3, ss="J", bit_depth=16, frames=5, fs=1024, seed=215, second=("stereo",), scene_second=True,
                                  gains=(-200, -600)),
    "toa_plus_scalable_C_s16": dict(order=3, ss="C", bit_depth=16, frames=8, fs=1024, seed=216, second=("scalable",), gains=(-400, -350)),
    "scalable_plus_foa_D_s24": dict(order=1, ss="D", bit_depth=24, frames=8, fs=1024, seed=217, second=("scalable",), scene_second=True,
                                    gains=(-500, -250)),
    "toa_plus_foa_B_s16": dict(order=3, ss="B", bit_depth=16, frames=6, fs=1024, seed=218, second=("scene", 1), gains=(-450, -300)),
    "foa_plus_toa_J_s16": dict(order=1, ss="J", bit_depth=16, frames=6, fs=1024, seed=219, second=("scene", 3), gains=(-350, -550)),
    "toa_projection_plus_soa_F_s16": dict(order=3, ss="F", bit_depth=16, frames=5, fs=1024, seed=220, projection=True,
                                          second=("scene", 2), gains=(-600, -400)),
}


def build_pair(name):
    """two-element presentations: element 1 = the case's scene-based element (mono or projection mode), element 2 =
    c["second"]; c["scene_second"] puts them into the presentation (and the stream) the other way round"""
    import e2e_cases as E
    c = E2E[name]
    fs, F, rate = c["fs"], c["frames"], 48000
    n = fs * F
    m = (c["order"] + 1) ** 2
    pd = lambda pid: W.param_definition(pid, rate, mode=1)
    s = W.sequence_header(1) + W.codec_config_lpcm(0, fs, 16, rate)

    def scene(eid, sid0, order, seed, projection):
        mm = (order + 1) ** 2
        xq = W.quantize(programme(seed, mm, n, rate), 16)
        if projection:
            subs_n, coupled = 10, 6
            rng = np.random.default_rng(seed)
            pq = rng.integers(-9000, 9000, size=(subs_n + coupled, mm)).astype(np.int16)
            pq[np.arange(mm), np.arange(mm)] = 29000
            desc = W.audio_element_ambisonics_projection(eid, 0, mm, list(range(sid0, sid0 + subs_n)), coupled, pq)

            def frame(f):
                subs, ch = [], 0
                for i in range(subs_n):
                    w = 2 if i < coupled else 1
                    subs.append((sid0 + i, W.lpcm_bytes(xq[ch:ch + w, f * fs:(f + 1) * fs], 16)))
                    ch += w
                return b"", subs
            return desc, frame, subs_n
        desc = W.audio_element_ambisonics_mono(eid, 0, mm, list(range(sid0, sid0 + mm)))
        return desc, (lambda f: (b"", [(sid0 + i, W.lpcm_bytes(xq[i:i + 1, f * fs:(f + 1) * fs], 16)) for i in range(mm)])), mm

    specs = [("scene", c["order"], c.get("projection", False)), c["second"]]
    if c.get("scene_second"):
        specs.reverse()
    descs, frames, sid = [], [], 0
    for k, sp in enumerate(specs):
        if sp[0] == "scene":
            d, fr, ns = scene(k + 1, sid, sp[1], c["seed"] + 10 * k, sp[2] if len(sp) > 2 else False)
        else:
            d, fr, _, ns = E._pair_element(sp[0], k + 1, sid, 200 + 10 * k, c["seed"] + 10 * k, n, fs, 16, rate, c)
        descs.append(d)
        frames.append(fr)
        sid += ns
    s += descs[0] + descs[1]
    s += W.mix_presentation(1, [dict(eid=1, pdef=pd(100), default_q78=c["gains"][0]), dict(eid=2, pdef=pd(102), default_q78=c["gains"][1])],
                            dict(pdef=pd(101), default_q78=0), [("ss", SS_ENUM[c["ss"]])])
    for f in range(F):
        ba, sa = frames[0](f)
        bb, sb = frames[1](f)
        s += W.temporal_delimiter() + ba + bb + W.audio_frames(sa + sb)
    return s, None


def build(name):
    c = E2E[name]
    if c.get("second"):
        return build_pair(name)
    fs, F, rate = c["fs"], c["frames"], c.get("rate", 48000)
    m = (c["order"] + 1) ** 2
    x = programme(c["seed"], m, fs * F, rate, c.get("silence_from"))
    xq = W.quantize(x, 16)
    pd = lambda pid: W.param_definition(pid, rate, mode=1)
    s = W.sequence_header(1) + W.codec_config_lpcm(0, fs, 16, rate)
    subs_n, coupled = 10, 6   # projection mode: 16 decoded channels in 10 sub-streams
    if c.get("projection"):
        rng = np.random.default_rng(c["seed"])
        pq = rng.integers(-9000, 9000, size=(subs_n + coupled, m)).astype(np.int16)
        pq[np.arange(m), np.arange(m)] = 29000
        s += W.audio_element_ambisonics_projection(1, 0, m, list(range(subs_n)), coupled, pq)
    else:
        s += W.audio_element_ambisonics_mono(1, 0, m, list(range(m)))
    s += W.mix_presentation(1, [dict(eid=1, pdef=pd(100), default_q78=0)], dict(pdef=pd(101), default_q78=0),
                            [("ss", SS_ENUM[c["ss"]])])
    for f in range(F):
        s += W.temporal_delimiter()
        if c.get("projection"):
            subs, ch = [], 0
            for i in range(subs_n):
                w = 2 if i < coupled else 1
                subs.append((i, W.lpcm_bytes(xq[ch:ch + w, f * fs:(f + 1) * fs], 16)))
                ch += w
            s += W.audio_frames(subs, trim=c.get("trims", {}).get(f))
        else:
            s += W.audio_frames([(i, W.lpcm_bytes(xq[i:i + 1, f * fs:(f + 1) * fs], 16)) for i in range(m)], trim=c.get("trims", {}).get(f))
    return s, xq
