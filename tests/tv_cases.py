"""End-to-end cases of the reference's -DSAMSUNG_TV build (the upstream default, CMakeLists.txt:14-19):
its own layout->layout tables (m2m_rdr.c:36-830), a 12-channel PCM stride whatever the layout
(IAMF_decoder.c:3492-3495), always the top layer of a scalable element (:1782-1822).  Shared by
oracle/gen_golden_tv.py (asks the real reference built that way) and tests/test_gpu_tv.py."""
import numpy as np

import e2e_cases as E
import iamf_writer as W
import synth

# name -> (builder, layout, bit depth, extra decode kwargs)
CASES = {
    "stereo_A_s16": dict(e2e="stereo_A_s16"),                 # BASELINE config 1 on the TV build
    "l714_J_s24_gain": dict(e2e="l714_J_s24_gain"),
    "l714_A_s16": dict(e2e="l714_A_s16"),
    "toa_H_s16": dict(e2e="toa_H_s16"),                       # 24 channels > 12: the surplus channels overwrite the next frame
    "two_elements_A_s32": dict(e2e="two_elements_A_s32"),
    "stereo_441_to_48k": dict(e2e="stereo_441_to_48k"),       # resampler in front of the 12-stride pack
    "scalable_C_s16": dict(e2e="scalable_C_s16"),             # TV: the 7.1.4 layer, not the matching 5.1.2 one
    "scalable_A_s16": dict(e2e="scalable_A_s16"),
    "scalable_B_s16": dict(e2e="scalable_B_s16"),
    "l714_J_ramps": dict(e2e="l714_J_ramps"),
    "l51_G_s16": dict(layout=("ss", 6), bit_depth=16, frames=4, fs=1024, seed=301, in_layout=2),   # 5.1 -> 14 ch (TV table differs)
    "l512_F_s16": dict(layout=("ss", 5), bit_depth=16, frames=4, fs=1024, seed=302, in_layout=3),  # 5.1.2 -> F
    "l312_J_s24": dict(layout=("ss", 9), bit_depth=24, frames=3, fs=960, seed=303, in_layout=8),   # 3.1.2 -> J
    # round 3: the e2e cases added for the default build, on the TV build's tables and stride
    "l51_A_s16": dict(e2e="l51_A_s16"),
    "l514_B_s24": dict(e2e="l514_B_s24"),
    "l71_D_s16": dict(e2e="l71_D_s16"),
    "l712_binaural_s16": dict(e2e="l712_binaural_s16"),
    "mono_A_s16": dict(e2e="mono_A_s16"),
    "l714_ext712_s16": dict(e2e="l714_ext712_s16"),
    "toa_G_s32": dict(e2e="toa_G_s32"),                       # 14 channels > 12
    "foa_binaural_s16": dict(e2e="foa_binaural_s16"),
    "stereo_fs128": dict(e2e="stereo_fs128"),
    "toa_H_fs2048": dict(e2e="toa_H_fs2048"),
    "scalable_0125_A": dict(e2e="scalable_0125_A"),           # TV: always the top layer (7.1), then 7.1 -> stereo
    "scalable_836_C": dict(e2e="scalable_836_C"),
}


def case(name):
    c = CASES[name]
    return E.CASES[c["e2e"]] if "e2e" in c else c


def build(name):
    c = CASES[name]
    if "e2e" in c:
        return E.build(c["e2e"])[0]
    fs, F, lay = c["fs"], c["frames"], c["in_layout"]
    ch = W.LAYOUT_CHANNELS[lay]
    x = np.clip(synth.hot(c["seed"], ch, fs * F, sigma=0.2, burst_amp=0.6, burst_phase=500, burst_period=3000),
                -1, 1 - 2 ** -15).astype(np.float32)
    desc, x_al, _ = E._channel_element(1, lay, x, 0, 16)
    pd = lambda pid: W.param_definition(pid, 48000, mode=1)
    s = W.sequence_header(1) + W.codec_config_lpcm(0, fs, 16, 48000) + desc
    s += W.mix_presentation(1, [dict(eid=1, pdef=pd(100), default_q78=0)], dict(pdef=pd(101), default_q78=0), [c["layout"]])
    for f in range(F):
        s += W.temporal_delimiter()
        s += W.audio_frames(W.channel_element_substreams(lay, x_al[:, f * fs:(f + 1) * fs], 0, 16))
    return s


def decode_kwargs(name):
    c = case(name)
    return dict(bit_depth=c.get("bit_depth", 16), out_rate=c.get("out_rate", 0), loudness=c.get("loudness", 0.0),
                limiter=c.get("limiter", True), threshold=c.get("threshold", -1.0), pcm_channels=12)


# Run-time output-layout switches of the -DSAMSUNG_TV build (IAMF_decoder.c:3819-3881): IAMF_decoder_output_layout_set_*
# + IAMF_decoder_configure(h, NULL, 0, NULL) between two frames.  name -> (stream case, layouts in turn, frames decoded
# before each switch)
SWITCH_CASES = {
    "switch_l714_J_to_A": dict(stream="l714_A_s16", layouts=[("ss", 9), ("ss", 0)], after=[3]),
    "switch_l714_A_to_J_to_B": dict(stream="l714_A_s16", layouts=[("ss", 0), ("ss", 9), ("ss", 1)], after=[2, 5]),
    "switch_toa_H_to_binaural": dict(stream="toa_H_s16", layouts=[("ss", 7), ("binaural",)], after=[2]),
    "switch_scalable_C_to_A": dict(stream="scalable_C_s16", layouts=[("ss", 2), ("ss", 0)], after=[3]),
    "switch_dmx_C_to_J": dict(stream="l714_C_dmx", layouts=[("ss", 2), ("ss", 9)], after=[2]),
    # both elements behind a stage of their own (a second batch for element 1): the demixers' and the down-mixers' states
    # across the switch, per element; a layout after which only one / none of the two stages remains
    "switch_2scalable_J_to_A_to_C": dict(stream="scalable_plus_scalable_J", layouts=[("ss", 9), ("ss", 0), ("ss", 2)], after=[3, 5]),
    "switch_2dmx_C_to_312_to_J": dict(stream="l714dmx_plus_l714dmx_C", layouts=[("ss", 2), ("ss", 11), ("ss", 9)], after=[2, 5]),
    "switch_scalable_dmx_312_to_B": dict(stream="scalable_plus_l714dmx_312", layouts=[("ss", 11), ("ss", 1)], after=[4]),
}
