"""Synthetic IAMF streams for the end-to-end goldens (oracle/gen_golden_extra.py runs the real
reference decoder on them) and for the tests that replay them through the oracle, the HIP path
and the IAMF_decoder.h facade.  build(name) -> (stream bytes, info) where info carries the
element PCM exactly as the renderer sees it (quantised LPCM values, playback channel order).
"""
import numpy as np

import iamf_writer as W
import synth

SS = dict(A=0, B=1, C=2, D=3, E=4, F=5, G=6, H=7, I=8, J=9, EXT712=10, EXT312=11, MONO=12)

# audio-layer order -> playback order (reference IAMF_utils.c:117-133 vs :181-196): playback
# channel p is audio-layer channel AL_OF_PLAYBACK[layout][p]
_CL = {1: "L2 R2", 2: "L5 R5 C LFE SL5 SR5", 7: "L7 R7 C LFE SL7 SR7 BL7 BR7 HFL HFR HBL HBR",
       3: "L5 R5 C LFE SL5 SR5 HL HR", 4: "L5 R5 C LFE SL5 SR5 HFL HFR HBL HBR", 0: "MONO",
       5: "L7 R7 C LFE SL7 SR7 BL7 BR7", 6: "L7 R7 C LFE SL7 SR7 BL7 BR7 HL HR", 8: "L3 R3 C LFE TL TR"}
_AL = {1: "L2 R2", 2: "L5 R5 SL5 SR5 C LFE", 7: "L7 R7 SL7 SR7 BL7 BR7 HFL HFR HBL HBR C LFE",
       3: "L5 R5 SL5 SR5 HL HR C LFE", 4: "L5 R5 SL5 SR5 HFL HFR HBL HBR C LFE", 0: "MONO",
       5: "L7 R7 SL7 SR7 BL7 BR7 C LFE", 6: "L7 R7 SL7 SR7 BL7 BR7 HL HR C LFE", 8: "L3 R3 TL TR C LFE"}


def al_index_of_playback(layout):
    cl, al = _CL[layout].split(), _AL[layout].split()
    return [al.index(c) for c in cl]


def _pdef_static(pid, rate=48000):
    return W.param_definition(pid, rate, mode=1)


def _descriptor_prefix(frame_size, sample_size=16, rate=48000, little_endian=True):
    return W.sequence_header(1) + W.codec_config_lpcm(0, frame_size, sample_size, rate, little_endian)


def _ss_layout(name):
    return ("ss", SS[name])


CASES = {
    # BASELINE configs[0]: stereo element -> Sound System A, 16 bit (the reference's own
    # CPU-runnable case, iamfplayer -o2 -s0)
    "stereo_A_s16": dict(layout=_ss_layout("A"), bit_depth=16, frames=20, fs=1024, seed=7),
    # frame sizes away from 1024 (the codec configuration's num_samples_per_frame is free): shorter than the limiter's
    # 240-sample delay (the first calls emit nothing or less than a frame), and longer than 1024 (max_frame_size = 6 frames,
    # IAMF_decoder.c:1628-1630)
    "stereo_fs128": dict(layout=_ss_layout("A"), bit_depth=16, frames=48, fs=128, seed=71),
    "stereo_fs2048": dict(layout=_ss_layout("A"), bit_depth=16, frames=4, fs=2048, seed=72),
    "toa_binaural_fs256": dict(layout=("binaural",), bit_depth=16, frames=24, fs=256, seed=1073),
    "toa_H_fs2048": dict(layout=_ss_layout("H"), bit_depth=16, frames=3, fs=2048, seed=74),
    # first- and second-order ambisonics (the order follows from the channel count, IAMF_decoder.c:2403-2413), and a limiter
    # threshold other than the default -1 dBFS
    "foa_binaural_s16": dict(layout=("binaural",), bit_depth=16, frames=6, fs=1024, seed=75, amb_ch=4),
    "soa_B_s24": dict(layout=_ss_layout("B"), bit_depth=24, frames=5, fs=1024, seed=76, amb_ch=9),
    "toa_binaural_thr6": dict(layout=("binaural",), bit_depth=16, frames=6, fs=1024, seed=77, threshold=-6.0),
    # the loudspeaker layouts the other cases do not touch: the permutation from audio-layer to playback order
    # (IAMF_utils.c:117-133 vs :181-196) and the layout -> layout matrix of each, end to end
    "mono_A_s16": dict(layout=_ss_layout("A"), bit_depth=16, frames=4, fs=1024, seed=81, ch_layout=0),
    "l51_A_s16": dict(layout=_ss_layout("A"), bit_depth=16, frames=4, fs=1024, seed=82, ch_layout=2),
    "l512_J_s16": dict(layout=_ss_layout("J"), bit_depth=16, frames=4, fs=1024, seed=83, ch_layout=3),
    "l514_B_s24": dict(layout=_ss_layout("B"), bit_depth=24, frames=4, fs=1024, seed=84, ch_layout=4),
    "l71_D_s16": dict(layout=_ss_layout("D"), bit_depth=16, frames=4, fs=1024, seed=85, ch_layout=5),
    "l712_binaural_s16": dict(layout=("binaural",), bit_depth=16, frames=4, fs=1024, seed=86, ch_layout=6),
    "l312_I_s16": dict(layout=_ss_layout("I"), bit_depth=16, frames=4, fs=1024, seed=87, ch_layout=8),
    "l514_312_s16": dict(layout=("ss", 11), bit_depth=16, frames=4, fs=1024, seed=88, ch_layout=4),
    # output sound systems the other cases do not reach (E, F, G, the 7.1.2 and mono extensions)
    "toa_G_s32": dict(layout=_ss_layout("G"), bit_depth=32, frames=4, fs=1024, seed=91),
    "toa_E_s16": dict(layout=_ss_layout("E"), bit_depth=16, frames=4, fs=1024, seed=92),
    "soa_F_s16": dict(layout=_ss_layout("F"), bit_depth=16, frames=4, fs=1024, seed=93, amb_ch=9),
    "foa_D_s16": dict(layout=_ss_layout("D"), bit_depth=16, frames=4, fs=1024, seed=94, amb_ch=4),
    "l714_ext712_s16": dict(layout=("ss", 10), bit_depth=16, frames=4, fs=1024, seed=95, ch_layout=7),
    "l714_mono_s16": dict(layout=("ss", 12), bit_depth=16, frames=4, fs=1024, seed=96, ch_layout=7),
    "toa_mono_s16": dict(layout=("ss", 12), bit_depth=16, frames=4, fs=1024, seed=97),
    # other stacks of scalable layers (demixer paths S1to2 / S2to3 / S3to5 / S5to7 / TF2toT2 / T2toT4 end to end), into an
    # output that takes the top layer and into one that selects a lower layer or the down-mixer
    "scalable_0125_I": dict(layout=_ss_layout("I"), bit_depth=16, frames=8, fs=1024, seed=101, scalable=True, layers=[0, 1, 2, 5]),
    "scalable_0125_A": dict(layout=_ss_layout("A"), bit_depth=16, frames=8, fs=1024, seed=101, scalable=True, layers=[0, 1, 2, 5]),
    "scalable_18_312": dict(layout=("ss", 11), bit_depth=16, frames=8, fs=1024, seed=102, scalable=True, layers=[1, 8]),
    "scalable_24_D": dict(layout=_ss_layout("D"), bit_depth=16, frames=8, fs=1024, seed=103, scalable=True, layers=[2, 4]),
    "scalable_34_binaural": dict(layout=("binaural",), bit_depth=16, frames=8, fs=1024, seed=104, scalable=True, layers=[3, 4]),
    "scalable_836_ext712": dict(layout=("ss", 10), bit_depth=16, frames=8, fs=1024, seed=105, scalable=True, layers=[8, 3, 6]),
    "scalable_836_C": dict(layout=_ss_layout("C"), bit_depth=16, frames=8, fs=1024, seed=105, scalable=True, layers=[8, 3, 6]),
    "toa_binaural_s16": dict(layout=("binaural",), bit_depth=16, frames=12, fs=1024, seed=1000),
    "toa_H_s16": dict(layout=_ss_layout("H"), bit_depth=16, frames=5, fs=1024, seed=13),
    "l714_J_s24_gain": dict(layout=_ss_layout("J"), bit_depth=24, frames=6, fs=960, seed=11,
                            element_gain_q78=-768, output_gain_q78=384),
    "two_elements_A_s32": dict(layout=_ss_layout("A"), bit_depth=32, frames=5, fs=1024, seed=21,
                               limiter=False, element_gain_q78=1536),
    "toa_binaural_loudness": dict(layout=("binaural",), bit_depth=16, frames=6, fs=1024, seed=31,
                                  loudness=-16.0, mix_loudness_q78=-24 * 256),
    "l714_A_s16": dict(layout=_ss_layout("A"), bit_depth=16, frames=5, fs=1024, seed=41),
    # parameter blocks: element mix gain (definition carries duration: one segment per block, step /
    # linear / Bezier in turn) and output mix gain (blocks carry two half-frame segments)
    "l714_J_ramps": dict(layout=_ss_layout("J"), bit_depth=16, frames=7, fs=1024, seed=51, ramps=True),
    # demixing parameter blocks -> parametric down-mixer 7.1.4 -> 5.1.2 (Sound System C)
    "l714_C_dmx": dict(layout=_ss_layout("C"), bit_depth=16, frames=8, fs=1024, seed=52,
                       dmx_modes=[1, 1, 2, 4, 5, 6, 0, 0], dmx_default=(1, 3)),
    # 44.1 kHz stream, 48 kHz output: speex-derived resampler in the path
    "stereo_441_to_48k": dict(layout=_ss_layout("A"), bit_depth=16, frames=6, fs=1024, seed=53, rate=44100,
                              out_rate=48000),
    # the other stream rates IAMF allows, to 48 kHz and (48 kHz stream) down to 44.1 kHz
    "stereo_96k_to_48k": dict(layout=_ss_layout("A"), bit_depth=16, frames=6, fs=1024, seed=58, rate=96000, out_rate=48000),
    "stereo_32k_to_48k": dict(layout=_ss_layout("A"), bit_depth=24, frames=6, fs=1024, seed=59, rate=32000, out_rate=48000),
    "stereo_16k_to_48k": dict(layout=_ss_layout("A"), bit_depth=16, frames=6, fs=1024, seed=60, rate=16000, out_rate=48000),
    "stereo_48k_to_441": dict(layout=_ss_layout("A"), bit_depth=16, frames=6, fs=1024, seed=66, rate=48000, out_rate=44100),
    # frame trimming: 100 samples off the first frame, 300 off the last
    "stereo_trim": dict(layout=_ss_layout("A"), bit_depth=16, frames=5, fs=1024, seed=54, trims={0: (100, 0), 4: (0, 300)}),
    # projection-mode ambisonics: 10 sub-streams (6 coupled) -> 16 decoded channels -> Q15 de-mapping
    # matrix -> 16 ambisonics channels (IAMF_core_decoder.c:116-130,228-252)
    "toa_projection_B_s16": dict(layout=_ss_layout("B"), bit_depth=16, frames=5, fs=1024, seed=55),
    # LPCM sample formats of the substreams (pcm/IAMF_pcm_decoder.c:60-83): 24 / 32 bit, big endian.
    # 24-bit big-endian goes through the reference's reads24be, which swaps the two upper bytes
    # (bitstream.c:204-208): the stream below decodes to what THAT produces
    "stereo_in24le": dict(layout=_ss_layout("A"), bit_depth=24, frames=4, fs=1024, seed=61, sample_size=24, lpcm=True),
    "stereo_in32le": dict(layout=_ss_layout("A"), bit_depth=32, frames=4, fs=1024, seed=62, sample_size=32, lpcm=True),
    "stereo_in16be": dict(layout=_ss_layout("A"), bit_depth=16, frames=4, fs=1024, seed=63, sample_size=16, lpcm=True,
                          big_endian=True),
    "stereo_in32be": dict(layout=_ss_layout("A"), bit_depth=16, frames=4, fs=1024, seed=64, sample_size=32, lpcm=True,
                          big_endian=True),
    "stereo_in24be_quirk": dict(layout=_ss_layout("A"), bit_depth=16, frames=4, fs=1024, seed=65, sample_size=24,
                                lpcm=True, big_endian=True),
    # scalable channel audio (N2): stereo -> 5.1.2 -> 7.1.4 with output gains on the first two layers,
    # recon-gain and demixing parameter blocks; the output layout selects the layer that is decoded
    # (IAMF_decoder.c:1776-1822), the demixer rebuilds the rest (demixer.c)
    # a reconfiguration must start from a clean database.  Three IA sequences in one file; the
    # first and third carry mix-gain parameter blocks on the SAME parameter ids (stale queues / timestamps
    # would shift or refuse the third one's ramps), the middle one has another frame count and trims
    "l714_A_ramps": dict(layout=_ss_layout("A"), bit_depth=16, frames=7, fs=1024, seed=57, ramps=True),
    "three_sequences_A_s16": dict(layout=_ss_layout("A"), bit_depth=16, concat=["l714_A_ramps", "stereo_trim", "l714_A_ramps"]),
    "scalable_J_s16": dict(layout=_ss_layout("J"), bit_depth=16, frames=8, fs=1024, seed=56, scalable=True),
    "scalable_C_s16": dict(layout=_ss_layout("C"), bit_depth=16, frames=8, fs=1024, seed=56, scalable=True),
    "scalable_A_s16": dict(layout=_ss_layout("A"), bit_depth=16, frames=8, fs=1024, seed=56, scalable=True),
    "scalable_binaural_s24": dict(layout=("binaural",), bit_depth=24, frames=8, fs=1024, seed=56, scalable=True),
    "scalable_B_s16": dict(layout=_ss_layout("B"), bit_depth=16, frames=8, fs=1024, seed=56, scalable=True),
    # 3.1.2 output: no layer matches -> 5.1.2 is decoded and demixed, then the parametric down-mixer
    # renders 5.1.2 -> 3.1.2 with the same demixing modes
    "scalable_312_dmx_s16": dict(layout=("ss", 11), bit_depth=16, frames=8, fs=1024, seed=56, scalable=True),
    # ---- round 4: presentations of TWO elements where one of them needs the per-stream stage the batch has
    # only once — the demixer of a scalable / output-gained element (IAMF_decoder.c:2351-2386), the parametric down-mixer
    # (:2448-2478), the projection de-mapping (IAMF_core_decoder.c:116-130) — in either position, mixed at :2702-2733
    "stereo_plus_scalable_J": dict(layout=_ss_layout("J"), bit_depth=16, frames=8, fs=1024, seed=201,
                                   pair=("stereo", "scalable"), element_gain_q78=-512, element2_gain_q78=-256),
    "scalable_plus_stereo_A": dict(layout=_ss_layout("A"), bit_depth=16, frames=8, fs=1024, seed=202,
                                   pair=("scalable", "stereo"), element_gain_q78=-300, element2_gain_q78=-700),
    "stereo_plus_scalable_C": dict(layout=_ss_layout("C"), bit_depth=24, frames=8, fs=1024, seed=203,
                                   pair=("stereo", "scalable"), element2_gain_q78=-384),
    "l714dmx_plus_stereo_C": dict(layout=_ss_layout("C"), bit_depth=16, frames=8, fs=1024, seed=204,
                                  pair=("l714dmx", "stereo"), element_gain_q78=-256, element2_gain_q78=-512,
                                  dmx_modes=[1, 1, 2, 4, 5, 6, 0, 0]),
    "stereo_plus_l714dmx_C": dict(layout=_ss_layout("C"), bit_depth=16, frames=8, fs=1024, seed=205,
                                  pair=("stereo", "l714dmx"), element_gain_q78=-512, element2_gain_q78=-256,
                                  dmx_modes=[2, 4, 5, 6, 0, 0, 1, 1]),
    "stereo_plus_l714dmx_312": dict(layout=("ss", 11), bit_depth=16, frames=8, fs=1024, seed=206,
                                    pair=("stereo", "l714dmx"), element2_gain_q78=-200, dmx_modes=[0, 1, 2, 4, 5, 6, 2, 1]),
    "stereo_plus_projection_B": dict(layout=_ss_layout("B"), bit_depth=16, frames=6, fs=1024, seed=207,
                                     pair=("stereo", "toa_projection"), element_gain_q78=-256, element2_gain_q78=-400),
    "l51_plus_projection_binaural": dict(layout=("binaural",), bit_depth=32, frames=6, fs=1024, seed=208,
                                         pair=("l51", "toa_projection"), element2_gain_q78=-300),
    # ---- BOTH elements need such a stage (the second one is rendered by a batch of its own into f32, the first one's batch
    # mixes it in as a plain second element: iamf_decoder_facade.c setup_pipeline)
    "scalable_plus_scalable_J": dict(layout=_ss_layout("J"), bit_depth=16, frames=8, fs=1024, seed=221,
                                     pair=("scalable", "scalable"), element_gain_q78=-400, element2_gain_q78=-650,
                                     scalable_modes2=[2, 0, 1, 6, 5, 4, 4, 1], recon_salt2=77),
    "scalable_plus_scalable_A": dict(layout=_ss_layout("A"), bit_depth=24, frames=8, fs=1024, seed=222,
                                     pair=("scalable", "scalable"), element_gain_q78=-300, element2_gain_q78=-500,
                                     scalable_modes2=[0, 2, 4, 6, 1, 5, 2, 0], recon_salt2=78),
    "scalable_plus_projection_B": dict(layout=_ss_layout("B"), bit_depth=16, frames=8, fs=1024, seed=223,
                                       pair=("scalable", "toa_projection"), element_gain_q78=-256, element2_gain_q78=-512),
    "projection_plus_scalable_C": dict(layout=_ss_layout("C"), bit_depth=16, frames=8, fs=1024, seed=224,
                                       pair=("toa_projection", "scalable"), element_gain_q78=-512, element2_gain_q78=-128),
    "projection_plus_projection_binaural": dict(layout=("binaural",), bit_depth=16, frames=6, fs=1024, seed=225,
                                                pair=("toa_projection", "toa_projection"), element_gain_q78=-700,
                                                element2_gain_q78=-900),
    "l714dmx_plus_l714dmx_C": dict(layout=_ss_layout("C"), bit_depth=16, frames=8, fs=1024, seed=226,
                                   pair=("l714dmx", "l714dmx"), element_gain_q78=-600, element2_gain_q78=-450,
                                   dmx_modes=[1, 1, 2, 4, 5, 6, 0, 0], dmx_modes2=[4, 5, 0, 0, 1, 2, 6, 6]),
    "l714dmx_plus_scalable_C": dict(layout=_ss_layout("C"), bit_depth=16, frames=8, fs=1024, seed=227,
                                    pair=("l714dmx", "scalable"), element_gain_q78=-500, element2_gain_q78=-350,
                                    dmx_modes=[6, 5, 4, 2, 1, 0, 0, 1], scalable_modes2=[1, 2, 2, 0, 6, 5, 4, 1]),
    "scalable_plus_l714dmx_312": dict(layout=("ss", 11), bit_depth=16, frames=8, fs=1024, seed=228,
                                      pair=("scalable", "l714dmx"), element_gain_q78=-450, element2_gain_q78=-550,
                                      dmx_modes=[0, 1, 2, 4, 5, 6, 2, 1]),
    # ... with animated mix gains on both elements and the output (element 1's ramp is applied by the second batch), with
    # trimmed first / last frames (both demixers start inside a frame), and behind the resampler
    "scalable_plus_scalable_J_ramps": dict(layout=_ss_layout("J"), bit_depth=16, frames=8, fs=1024, seed=231,
                                           pair=("scalable", "scalable"), pair_ramps=True, scalable_modes2=[2, 0, 1, 6, 5, 4, 4, 1],
                                           recon_salt2=79),
    "stereo_plus_scalable_C_ramps": dict(layout=_ss_layout("C"), bit_depth=16, frames=8, fs=1024, seed=232,
                                         pair=("stereo", "scalable"), pair_ramps=True),
    "l714dmx_plus_l714dmx_C_trim": dict(layout=_ss_layout("C"), bit_depth=16, frames=8, fs=1024, seed=233,
                                        pair=("l714dmx", "l714dmx"), element_gain_q78=-300, element2_gain_q78=-420,
                                        dmx_modes=[1, 2, 4, 5, 6, 0, 0, 1], dmx_modes2=[6, 5, 4, 0, 1, 2, 2, 0],
                                        trims={0: (100, 0), 7: (0, 300)}),
    "scalable_plus_projection_B_trim": dict(layout=_ss_layout("B"), bit_depth=24, frames=8, fs=1024, seed=234,
                                            pair=("scalable", "toa_projection"), element_gain_q78=-256, element2_gain_q78=-512,
                                            trims={0: (260, 0), 3: (0, 0), 7: (0, 41)}),
    "scalable_plus_scalable_A_441_to_48k": dict(layout=_ss_layout("A"), bit_depth=16, frames=8, fs=1024, seed=235, rate=44100,
                                                out_rate=48000, pair=("scalable", "scalable"), element_gain_q78=-350,
                                                element2_gain_q78=-450, recon_salt2=80),
    # a presentation that says more about itself (two layouts, true peak, anchored loudness): what
    # IAMF_decoder_get_last_metadata hands out (IAMF_decoder.c:3619-3706)
    "stereo_loudness_info": dict(layout=_ss_layout("A"), bit_depth=16, frames=4, fs=1024, seed=209, loudness_infos=True),
}

PLAIN_LAYOUT_KINDS = dict(mono=0, stereo=1, l51=2, l512=3, l514=4, l71=5, l712=6, l714=7, l312=8)
SCALABLE_LAYERS = [1, 3, 7]
SCALABLE_GAINS = {0: (0b110000, -768), 1: (0b001111, 384)}   # layer -> (flags, q7.8 dB)
SCALABLE_MODES = [1, 1, 2, 4, 5, 6, 0, 2]


def scalable_recon_bytes(frame, n, salt=0):
    rng = np.random.default_rng(5600 + frame + 1000 * salt)
    return [int(v) for v in rng.integers(100, 256, size=n)]


def _channel_element(eid, layout, x_playback, first_sid, sample_size):
    """returns (descriptor obu, function frame_index -> substreams, quantised playback PCM)"""
    xq = W.quantize(x_playback, sample_size)
    perm = al_index_of_playback(layout)
    x_al = np.empty_like(xq)
    for p, a in enumerate(perm):
        x_al[a] = xq[p]
    ns = W.LAYOUT_SUBSTREAMS[layout][0]
    desc = W.audio_element_channel(eid, 0, layout, list(range(first_sid, first_sid + ns)))
    return desc, x_al, xq


def _toa_element(eid, x, first_sid, sample_size):
    xq = W.quantize(x, sample_size)
    desc = W.audio_element_ambisonics_mono(eid, 0, x.shape[0], list(range(first_sid, first_sid + x.shape[0])))
    return desc, xq


def _block_keep(c, pid0):
    """c["drop_blocks"] = (seed, probability): the demixing (k = 0) / recon-gain (k = 1) block of a frame is left out with
    that probability (the decoder then goes on with the mode / the gains it has)"""
    if not c.get("drop_blocks"):
        return lambda f, k: True
    sd, prob = c["drop_blocks"]
    return lambda f, k: np.random.default_rng(sd * 1000003 + pid0 * 131 + f * 2 + k).random() >= prob


# ---- one element of a two-element presentation: descriptor, per-frame (parameter blocks, sub-streams), what the renderer sees ----
def _pair_element(kind, eid, sid0, pid0, seed, n, fs, ss, rate, c):
    """returns (descriptor obus, frame -> (parameter block obus, [(sub-stream id, bytes)]), info element, sub-streams used)"""
    second = eid == 2   # the second element's own schedules, where the case names them
    dmx_modes = c.get("dmx_modes2" if second else "dmx_modes", c.get("dmx_modes"))
    sc_modes = c.get("scalable_modes2", SCALABLE_MODES) if second else c.get("scalable_modes1", SCALABLE_MODES)
    salt = c.get("recon_salt2", 0) if second else 0
    if kind in ("zoa", "foa", "soa", "toa"):   # mono-coded ambisonics, one sub-stream per channel
        order = ("zoa", "foa", "soa", "toa").index(kind)
        m = (order + 1) ** 2
        x = np.clip(synth.hot(seed, m, n, sigma=0.16, burst_amp=0.5, burst_phase=450, burst_period=2900), -1, 1 - 2 ** -15).astype(np.float32)
        xq = W.quantize(x, ss)
        desc = W.audio_element_ambisonics_mono(eid, 0, m, list(range(sid0, sid0 + m)))
        return (desc, lambda f: (b"", [(sid0 + i, W.lpcm_bytes(xq[i:i + 1, f * fs:(f + 1) * fs], ss)) for i in range(m)]),
                dict(kind="scene", order=order, x=xq), m)
    if kind in PLAIN_LAYOUT_KINDS:
        lay = PLAIN_LAYOUT_KINDS[kind]
        x = np.clip(synth.hot(seed, W.LAYOUT_CHANNELS[lay], n, sigma=0.18, burst_amp=0.5, burst_phase=350, burst_period=3100),
                    -1, 1 - 2 ** -15).astype(np.float32)
        desc, x_al, xq = _channel_element(eid, lay, x, sid0, ss)
        return (desc, lambda f: (b"", W.channel_element_substreams(lay, x_al[:, f * fs:(f + 1) * fs], sid0, ss)),
                dict(kind="channel", layout=lay, x=xq), W.LAYOUT_SUBSTREAMS[lay][0])
    if kind == "l714dmx" or kind.startswith("dmx:"):   # a channel-based element with demixing info ("dmx:<layout id>")
        lay = 7 if kind == "l714dmx" else int(kind[4:])
        nch = W.LAYOUT_CHANNELS[lay]
        x = np.clip(synth.hot(seed, nch, n, sigma=0.18, burst_amp=0.5, burst_phase=500, burst_period=3000),
                    -1, 1 - 2 ** -15).astype(np.float32)
        xq = W.quantize(x, ss)
        x_al = np.empty_like(xq)
        for p_, a_ in enumerate(al_index_of_playback(lay)):
            x_al[a_] = xq[p_]
        dmode, dw = c.get("dmx_default2" if second else "dmx_default1", (1, 3))
        nsub = W.LAYOUT_SUBSTREAMS[lay][0]
        desc = W.audio_element_channel(eid, 0, lay, list(range(sid0, sid0 + nsub)),
                                       demixing=dict(pid=pid0, rate=rate, frame=fs, mode=dmode, w=dw))
        keep = _block_keep(c, pid0)
        return (desc, lambda f: (W.demixing_block(pid0, dmx_modes[f]) if keep(f, 0) else b"",
                                 W.channel_element_substreams(lay, x_al[:, f * fs:(f + 1) * fs], sid0, ss)),
                dict(kind="channel", layout=lay, x=xq), nsub)
    if kind == "scalable":
        import demix_cases as D
        layers = c.get("scalable_layers2" if second else "scalable_layers1", SCALABLE_LAYERS)
        lgains = c.get("scalable_gains2" if second else "scalable_gains1", SCALABLE_GAINS)
        dmode, dw = c.get("dmx_default2" if second else "dmx_default1", (1, 3))
        order, per_layer = D.channels_order(layers)
        xd = W.quantize(synth.hot(seed, len(order), n, sigma=0.13, burst_amp=0.4, burst_phase=600,
                                  burst_period=2700).clip(-1, 1 - 2 ** -15).astype(np.float32), ss)
        wl = []
        for li, (lay, pl) in enumerate(zip(layers, per_layer)):
            rf = D.recon_flags(layers[0], lay) if li else 0
            wl.append(dict(layout=lay, nsub=pl["substreams"], ncoupled=pl["coupled"],
                           out_gain=lgains.get(li), recon=bool(rf), recon_flags=rf))
        nsub = sum(l["nsub"] for l in wl)
        desc = W.audio_element_scalable(eid, 0, wl, list(range(sid0, sid0 + nsub)),
                                        demixing=dict(pid=pid0, rate=rate, frame=fs, mode=dmode, w=dw),
                                        recon=dict(pid=pid0 + 1, rate=rate, frame=fs))

        keep = _block_keep(c, pid0)

        def frame(f):
            blocks = W.demixing_block(pid0, sc_modes[f]) if keep(f, 0) else b""
            if keep(f, 1):
                blocks += W.recon_gain_block(pid0 + 1, [(l["recon_flags"], scalable_recon_bytes(f, bin(l["recon_flags"]).count("1"), salt))
                                                        for l in wl if l["recon"]])
            subs, ch, sid = [], 0, sid0
            for l in wl:
                for k in range(l["nsub"]):
                    w = 2 if k < l["ncoupled"] else 1
                    subs.append((sid, W.lpcm_bytes(xd[ch:ch + w, f * fs:(f + 1) * fs], ss)))
                    ch += w
                    sid += 1
            return blocks, subs
        return desc, frame, dict(kind="scalable", layers=layers, order=order, x=xd, wl=wl, gains=lgains, modes=sc_modes,
                                 salt=salt), nsub
    if kind == "toa_projection":
        subs_n, coupled = 10, 6
        rng = np.random.default_rng(seed)
        pq = rng.integers(-9000, 9000, size=(subs_n + coupled, 16)).astype(np.int16)
        pq[np.arange(16), np.arange(16)] = 29000
        xd = W.quantize(synth.hot(seed, 16, n, sigma=0.1, burst_amp=0.4, burst_phase=700, burst_period=2500)
                        .clip(-1, 1 - 2 ** -15).astype(np.float32), ss)
        desc = W.audio_element_ambisonics_projection(eid, 0, 16, list(range(sid0, sid0 + subs_n)), coupled, pq)
        pf = pq.astype(np.float32) * np.float32(2.0 ** -15)
        xa = np.zeros((16, n), np.float32)
        for l in range(subs_n + coupled):
            xa = (xa + (xd[l][None, :] * pf[l][:, None]).astype(np.float32)).astype(np.float32)

        def frame(f):
            subs, ch = [], 0
            for i in range(subs_n):
                w = 2 if i < coupled else 1
                subs.append((sid0 + i, W.lpcm_bytes(xd[ch:ch + w, f * fs:(f + 1) * fs], ss)))
                ch += w
            return b"", subs
        return desc, frame, dict(kind="scene", order=3, x=xa), subs_n
    raise KeyError(kind)


def build(name):
    c = CASES[name]
    if c.get("concat"):   # several IA sequences back to back: the decoder must be reconfigured at each header
        parts = [build(p) for p in c["concat"]]
        return b"".join(p[0] for p in parts), dict(case=c, elements=[], parts=[p[1] for p in parts])
    fs, F = c["fs"], c["frames"]
    n = fs * F
    ss = c.get("sample_size", 16)
    rate = c.get("rate", 48000)
    le = not c.get("big_endian", False)
    stream = _descriptor_prefix(fs, ss, rate, le)
    info = dict(case=c, elements=[])
    eg = c.get("element_gain_q78", 0)
    og = c.get("output_gain_q78", 0)
    lay = [c["layout"]] if c["layout"][0] == "ss" else [("binaural",)]
    layouts_field = [("ss", c["layout"][1])] if c["layout"][0] == "ss" else [("binaural",)]

    def frames_of(subs_fn):
        return subs_fn

    if c.get("pair"):
        W.LE_DEFAULT = le   # (restored below: every sub-stream of this stream in the codec configuration's byte order)
        ka = c["pair"][0]
        kb = c["pair"][1] if len(c["pair"]) > 1 else None
        da, fa, ia, na = _pair_element(ka, 1, 0, 200, c["seed"], n, fs, ss, rate, c)
        stream += da
        gs = c.get("gain_sched", {})   # pid -> dict(pdef, blocks[frame]): a case's own mix-gain parameter timelines
        pdef_of = lambda pid: gs[pid]["pdef"] if pid in gs else _pdef_static(pid, rate)
        els = [dict(eid=1, pdef=pdef_of(100), default_q78=eg)]
        if kb:
            db, fb, ib, nb = _pair_element(kb, 2, na, 210, c["seed"] + 1, n, fs, ss, rate, c)
            stream += db
            els.append(dict(eid=2, pdef=pdef_of(102), default_q78=c.get("element2_gain_q78", 0)))
        else:
            fb, ib = (lambda f: (b"", [])), None
        stream += W.mix_presentation(1, els, dict(pdef=pdef_of(101), default_q78=og), layouts_field,
                                     loudness_q78=c.get("mix_loudness_q78", 0))
        info["elements"] += [ia, ib] if kb else [ia]
        m1 = dict(duration=fs, constant_interval=fs)
        for f in range(F):
            ba, sa = fa(f)
            bb, sb = fb(f)
            stream += W.temporal_delimiter()
            if c.get("pair_ramps"):   # mix-gain blocks of both elements and (from the second frame on) of the output
                a0, b0 = -64 * f, -700 + 90 * f
                stream += W.mix_gain_block(100, [dict(anim=W.ANIM_LINEAR, start=a0, end=a0 - 64)], mode1=m1)
                if kb:
                    stream += W.mix_gain_block(102, [dict(anim=W.ANIM_BEZIER, start=b0, end=b0 + 90, control=b0 + 200, rel_time=(40 + 20 * f) % 256)
                                                     if f % 3 != 2 else dict(anim=W.ANIM_STEP, start=b0)], mode1=m1)
                if f >= 1:
                    stream += W.mix_gain_block(101, [dict(anim=W.ANIM_LINEAR, start=100 - 40 * f, end=60 - 40 * f),
                                                     dict(anim=W.ANIM_STEP, start=60 - 40 * f)],
                                               mode1=dict(duration=fs, constant_interval=0, intervals=[fs // 4, fs - fs // 4]))
            for pid in sorted(gs):
                if pid != 102 or kb:
                    stream += gs[pid]["blocks"][f]
            stream += ba + bb + W.audio_frames(sa + sb, trim=c.get("trims", {}).get(f))
        W.LE_DEFAULT = True
        return stream, info
    if name == "stereo_loudness_info":
        x = synth.uniform(c["seed"], 2, n, 0.7)
        desc, x_al, xq = _channel_element(1, 1, x, 0, ss)
        stream += desc
        stream += W.mix_presentation(1, [dict(eid=1, pdef=_pdef_static(100, rate), default_q78=eg)],
                                     dict(pdef=_pdef_static(101, rate), default_q78=og), [("ss", 0), ("binaural",), ("ss", 9)],
                                     loudness_infos=[dict(integrated=-6144, peak=-512, true_peak=-300, anchors=[(1, -5888), (2, -6400)]),
                                                     dict(integrated=-5632, peak=-256),
                                                     dict(integrated=-6400, peak=-700, anchors=[(0, -6100)])])
        info["elements"].append(dict(kind="channel", layout=1, x=xq))
        for f in range(F):
            stream += W.temporal_delimiter()
            stream += W.audio_frames(W.channel_element_substreams(1, x_al[:, f * fs:(f + 1) * fs], 0, ss))
        return stream, info
    if name in ("stereo_A_s16", "stereo_441_to_48k", "stereo_trim", "stereo_fs128", "stereo_fs2048", "stereo_96k_to_48k",
                "stereo_32k_to_48k", "stereo_16k_to_48k", "stereo_48k_to_441"):
        x = synth.uniform(c["seed"], 2, n, 0.9)
        desc, x_al, xq = _channel_element(1, 1, x, 0, ss)
        stream += desc
        stream += W.mix_presentation(1, [dict(eid=1, pdef=_pdef_static(100, rate), default_q78=eg)],
                                     dict(pdef=_pdef_static(101, rate), default_q78=og), layouts_field)
        info["elements"].append(dict(kind="channel", layout=1, x=xq))
        for f in range(F):
            stream += W.temporal_delimiter()
            stream += W.audio_frames(W.channel_element_substreams(1, x_al[:, f * fs:(f + 1) * fs], 0, ss),
                                     trim=c.get("trims", {}).get(f))
    elif name in ("l714_J_ramps", "l714_A_ramps"):
        x = np.clip(synth.hot(c["seed"], 12, n, sigma=0.2, burst_amp=0.6, burst_phase=500, burst_period=3000),
                    -1, 1 - 2 ** -15).astype(np.float32)
        desc, x_al, xq = _channel_element(1, 7, x, 0, ss)
        stream += desc
        el_def = W.param_definition(100, 48000, mode=0, duration=fs, constant_interval=fs)
        out_def = W.param_definition(101, 48000, mode=1)
        stream += W.mix_presentation(1, [dict(eid=1, pdef=el_def, default_q78=-256)],
                                     dict(pdef=out_def, default_q78=128), layouts_field)
        info["elements"].append(dict(kind="channel", layout=7, x=xq))
        el_blocks = [dict(anim=W.ANIM_STEP, start=-512), dict(anim=W.ANIM_LINEAR, start=-512, end=256),
                     dict(anim=W.ANIM_BEZIER, start=256, end=-768, control=-128, rel_time=64),
                     dict(anim=W.ANIM_LINEAR, start=-768, end=0), dict(anim=W.ANIM_STEP, start=0),
                     dict(anim=W.ANIM_BEZIER, start=0, end=-300, control=200, rel_time=192),
                     dict(anim=W.ANIM_STEP, start=-300)]
        info["el_blocks"] = el_blocks
        for f in range(F):
            stream += W.temporal_delimiter()
            stream += W.mix_gain_block(100, [el_blocks[f]])
            if f >= 1:  # the first frame runs on the default output gain
                stream += W.mix_gain_block(101, [dict(anim=W.ANIM_LINEAR, start=128 - 64 * f, end=128 - 64 * f - 32),
                                                 dict(anim=W.ANIM_STEP, start=128 - 64 * f - 32)],
                                           mode1=dict(duration=fs, constant_interval=0, intervals=[fs // 2, fs // 2]))
            stream += W.audio_frames(W.channel_element_substreams(7, x_al[:, f * fs:(f + 1) * fs], 0, ss))
    elif name == "l714_C_dmx":
        x = np.clip(synth.hot(c["seed"], 12, n, sigma=0.2, burst_amp=0.6, burst_phase=500, burst_period=3000),
                    -1, 1 - 2 ** -15).astype(np.float32)
        xq = W.quantize(x, ss)
        perm = al_index_of_playback(7)
        x_al = np.empty_like(xq)
        for p_, a_ in enumerate(perm):
            x_al[a_] = xq[p_]
        dm, dw = c["dmx_default"]
        stream += W.audio_element_channel(1, 0, 7, list(range(7)),
                                          demixing=dict(pid=200, rate=48000, frame=fs, mode=dm, w=dw))
        stream += W.mix_presentation(1, [dict(eid=1, pdef=_pdef_static(100), default_q78=eg)],
                                     dict(pdef=_pdef_static(101), default_q78=og), layouts_field)
        info["elements"].append(dict(kind="channel", layout=7, x=xq))
        for f in range(F):
            stream += W.temporal_delimiter()
            stream += W.demixing_block(200, c["dmx_modes"][f])
            stream += W.audio_frames(W.channel_element_substreams(7, x_al[:, f * fs:(f + 1) * fs], 0, ss))
    elif name in ("toa_binaural_s16", "toa_H_s16", "toa_binaural_loudness", "toa_binaural_fs256", "toa_H_fs2048",
                  "foa_binaural_s16", "soa_B_s24", "toa_binaural_thr6", "toa_G_s32", "toa_E_s16", "soa_F_s16", "foa_D_s16",
                  "toa_mono_s16"):
        ach = c.get("amb_ch", 16)
        if name in ("toa_H_s16", "toa_H_fs2048"):
            x = synth.gaussian(c["seed"], ach, n, 0.15)
        else:
            x = np.clip(synth.hot(c["seed"], ach, n, sigma=0.2, burst_amp=0.7, burst_phase=900, burst_period=5000),
                        -1, 1 - 2 ** -15).astype(np.float32)
        desc, xq = _toa_element(1, x, 0, ss)
        stream += desc
        stream += W.mix_presentation(1, [dict(eid=1, pdef=_pdef_static(100), default_q78=eg)],
                                     dict(pdef=_pdef_static(101), default_q78=og), layouts_field,
                                     loudness_q78=c.get("mix_loudness_q78", 0))
        info["elements"].append(dict(kind="scene", order={4: 1, 9: 2, 16: 3}[ach], x=xq))
        for f in range(F):
            stream += W.temporal_delimiter()
            subs = [(i, W.lpcm_bytes(xq[i:i + 1, f * fs:(f + 1) * fs], ss)) for i in range(ach)]
            stream += W.audio_frames(subs)
    elif c.get("lpcm"):
        x = synth.uniform(c["seed"], 2, n, 0.6)
        xq = W.quantize(x, ss)
        if ss == 24 and not le:   # what reads24be makes of big-endian bytes b0 b1 b2: (b1 << 16) | (b0 << 8) | b2
            v = np.round(x.astype(np.float64) * 8388608.0).clip(-2 ** 23, 2 ** 23 - 1).astype(np.int64) & 0xffffff
            b0, b1, b2 = v >> 16, (v >> 8) & 0xff, v & 0xff
            u = (b1 << 16) | (b0 << 8) | b2
            u = np.where(u & 0x800000, u - (1 << 24), u)
            xq = (u.astype(np.float32) / np.float32(8388608.0)).astype(np.float32)
        stream += W.audio_element_channel(1, 0, 1, [0])
        stream += W.mix_presentation(1, [dict(eid=1, pdef=_pdef_static(100), default_q78=eg)],
                                     dict(pdef=_pdef_static(101), default_q78=og), layouts_field)
        info["elements"].append(dict(kind="channel", layout=1, x=xq))
        for f in range(F):
            stream += W.temporal_delimiter()
            stream += W.audio_frames([(0, W.lpcm_bytes(x[:, f * fs:(f + 1) * fs], ss, le))])
    elif c.get("scalable"):
        import demix_cases as D
        layers = c.get("layers", SCALABLE_LAYERS)
        lgains = SCALABLE_GAINS if layers == SCALABLE_LAYERS else {}
        order, per_layer = D.channels_order(layers)
        xd = W.quantize(synth.hot(c["seed"], len(order), n, sigma=0.15, burst_amp=0.45, burst_phase=600,
                                  burst_period=2700).clip(-1, 1 - 2 ** -15).astype(np.float32), ss)
        wl, sid = [], 0
        for li, (lay, pl) in enumerate(zip(layers, per_layer)):
            # recon gains ride on the layers above the first: flags = what that layer needs rebuilt
            rf = D.recon_flags(layers[0], lay) if li else 0
            wl.append(dict(layout=lay, nsub=pl["substreams"], ncoupled=pl["coupled"],
                           out_gain=lgains.get(li), recon=bool(rf), recon_flags=rf))
        nsub = sum(l["nsub"] for l in wl)
        stream += W.audio_element_scalable(1, 0, wl, list(range(nsub)),
                                           demixing=dict(pid=200, rate=rate, frame=fs, mode=1, w=3),
                                           recon=dict(pid=201, rate=rate, frame=fs))
        stream += W.mix_presentation(1, [dict(eid=1, pdef=_pdef_static(100), default_q78=eg)],
                                     dict(pdef=_pdef_static(101), default_q78=og), layouts_field)
        info["elements"].append(dict(kind="scalable", layers=layers, order=order, x=xd, wl=wl, gains=lgains))
        for f in range(F):
            stream += W.temporal_delimiter()
            stream += W.demixing_block(200, SCALABLE_MODES[f])
            if True:   # a block per frame: the reference's parameter timeline does not survive gaps
                stream += W.recon_gain_block(201, [(l["recon_flags"], scalable_recon_bytes(f, bin(l["recon_flags"]).count("1")))
                                                   for l in wl if l["recon"]])
            subs, ch, sid = [], 0, 0
            for l in wl:
                for k in range(l["nsub"]):
                    w = 2 if k < l["ncoupled"] else 1
                    subs.append((sid, W.lpcm_bytes(xd[ch:ch + w, f * fs:(f + 1) * fs], ss)))
                    ch += w
                    sid += 1
            stream += W.audio_frames(subs)
    elif name == "toa_projection_B_s16":
        subs_n, coupled = 10, 6
        rng = np.random.default_rng(c["seed"])
        pq = rng.integers(-9000, 9000, size=(subs_n + coupled, 16)).astype(np.int16)
        pq[np.arange(16), np.arange(16)] = 29000
        xd = W.quantize(synth.hot(c["seed"], 16, n, sigma=0.12, burst_amp=0.5, burst_phase=700, burst_period=2500)
                        .clip(-1, 1 - 2 ** -15).astype(np.float32), ss)   # decoded channels
        stream += W.audio_element_ambisonics_projection(1, 0, 16, list(range(subs_n)), coupled, pq)
        stream += W.mix_presentation(1, [dict(eid=1, pdef=_pdef_static(100), default_q78=eg)],
                                     dict(pdef=_pdef_static(101), default_q78=og), layouts_field)
        pf = pq.astype(np.float32) * np.float32(2.0 ** -15)
        xa = np.zeros((16, n), np.float32)
        for l in range(subs_n + coupled):
            xa = (xa + (xd[l][None, :] * pf[l][:, None]).astype(np.float32)).astype(np.float32)
        info["elements"].append(dict(kind="scene", order=3, x=xa))
        for f in range(F):
            stream += W.temporal_delimiter()
            subs, ch = [], 0
            for i in range(subs_n):
                w = 2 if i < coupled else 1
                subs.append((i, W.lpcm_bytes(xd[ch:ch + w, f * fs:(f + 1) * fs], ss)))
                ch += w
            stream += W.audio_frames(subs)
    elif c.get("ch_layout") is not None:
        lay_id = c["ch_layout"]
        nch = W.LAYOUT_CHANNELS[lay_id]
        x = np.clip(synth.hot(c["seed"], nch, n, sigma=0.2, burst_amp=0.6, burst_phase=400, burst_period=2900),
                    -1, 1 - 2 ** -15).astype(np.float32)
        desc, x_al, xq = _channel_element(1, lay_id, x, 0, ss)
        stream += desc
        stream += W.mix_presentation(1, [dict(eid=1, pdef=_pdef_static(100), default_q78=eg)],
                                     dict(pdef=_pdef_static(101), default_q78=og), layouts_field)
        info["elements"].append(dict(kind="channel", layout=lay_id, x=xq))
        for f in range(F):
            stream += W.temporal_delimiter()
            stream += W.audio_frames(W.channel_element_substreams(lay_id, x_al[:, f * fs:(f + 1) * fs], 0, ss))
    elif name in ("l714_J_s24_gain", "l714_A_s16"):
        x = np.clip(synth.hot(c["seed"], 12, n, sigma=0.2, burst_amp=0.6, burst_phase=500, burst_period=3000),
                    -1, 1 - 2 ** -15).astype(np.float32)
        desc, x_al, xq = _channel_element(1, 7, x, 0, ss)
        stream += desc
        stream += W.mix_presentation(1, [dict(eid=1, pdef=_pdef_static(100), default_q78=eg)],
                                     dict(pdef=_pdef_static(101), default_q78=og), layouts_field)
        info["elements"].append(dict(kind="channel", layout=7, x=xq))
        for f in range(F):
            stream += W.temporal_delimiter()
            stream += W.audio_frames(W.channel_element_substreams(7, x_al[:, f * fs:(f + 1) * fs], 0, ss))
    elif name == "two_elements_A_s32":
        xs = synth.uniform(c["seed"], 2, n, 0.9)
        xs[0, 100] = 0.99997
        xs[1, 101] = -1.0
        xt = synth.gaussian(c["seed"] + 1, 16, n, 0.15)
        d1, xs_al, xsq = _channel_element(1, 1, xs, 0, ss)
        d2, xtq = _toa_element(2, xt, 1, ss)
        stream += d1 + d2
        stream += W.mix_presentation(1, [dict(eid=1, pdef=_pdef_static(100), default_q78=eg),
                                         dict(eid=2, pdef=_pdef_static(102), default_q78=0)],
                                     dict(pdef=_pdef_static(101), default_q78=og), layouts_field)
        info["elements"].append(dict(kind="channel", layout=1, x=xsq))
        info["elements"].append(dict(kind="scene", order=3, x=xtq))
        for f in range(F):
            stream += W.temporal_delimiter()
            subs = W.channel_element_substreams(1, xs_al[:, f * fs:(f + 1) * fs], 0, ss)
            subs += [(1 + i, W.lpcm_bytes(xtq[i:i + 1, f * fs:(f + 1) * fs], ss)) for i in range(16)]
            stream += W.audio_frames(subs)
    else:
        raise KeyError(name)
    return stream, info


# What IAMF_decoder_get_last_metadata reports while these streams decode (tests/golden/meta.npz, rows by
# decoder_driver.last_metadata from the REAL reference): pts arithmetic with another time base and a start offset, a
# re-based clock mid-stream, trims, the resampler's rates, binaural / multichannel / mixed sound modes, loudness records
# with true peak and anchors, the DEMIXING record and its mode per frame.
META_CASES = {
    "stereo_A_s16": dict(pts=(0, 90000)),
    "stereo_loudness_info": dict(pts=(123456, 44100), set_pts_after=2, set_pts_to=(777, 1000)),
    "stereo_trim": dict(pts=(1000, 90000)),
    "stereo_441_to_48k": dict(pts=(5, 90000)),
    "stereo_48k_to_441": dict(pts=(0, 48000)),
    "toa_binaural_s16": dict(pts=(0, 90000)),
    "two_elements_A_s32": dict(pts=(0, 1000)),
    "l714_C_dmx": dict(pts=(0, 90000)),
    "scalable_A_s16": dict(pts=(0, 90000)),
    "scalable_J_s16": dict(pts=(0, 90000)),
    "stereo_plus_scalable_J": dict(pts=(0, 90000)),
    "stereo_plus_l714dmx_C": dict(pts=(0, 90000)),
    "l714dmx_plus_stereo_C": dict(pts=(0, 90000)),
    "l51_plus_projection_binaural": dict(pts=(0, 90000)),
    "stereo_fs128": dict(pts=(0, 90000)),
    "l714dmx_plus_l714dmx_C": dict(pts=(0, 90000)),
    "l714dmx_plus_scalable_C": dict(pts=(0, 48000)),
    "scalable_plus_l714dmx_312": dict(pts=(17, 90000)),
}
