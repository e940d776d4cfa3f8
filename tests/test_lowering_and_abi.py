"""Readers, lowering, blob layout and the C-ABI surface — all on CPU (no compute calls without a GPU)."""

import ctypes
import os
import re
import struct

import numpy as np
import pytest

from conftest import KERAS_PATH, PKG, REPO, TFLITE_PATH


def _plans(keep_all=False, fuse=False):
    from birdnet_stm32.models.runners import lower_model_file

    return lower_model_file(KERAS_PATH, keep_all=keep_all, fuse=fuse), lower_model_file(TFLITE_PATH, keep_all=keep_all, fuse=fuse)


def test_fused_plan_structure():
    """The production plans: front block + fused depthwise/pointwise blocks on the matrix cores, audio-path operators."""
    from birdnet_stm32.models import _pack as pk

    f32, i8 = _plans(fuse=True)
    kinds = [pk.KIND_NAMES[o.kind] for o in f32.ops]
    assert kinds == ["f32_mel", "f32_stftmel", "f32_front", "f32_front"] + ["f32_dwpw"] * 10 + ["f32_gapdense"]
    paths = [o.p[pk.OP_PATH] for o in f32.ops]
    assert paths[:4] == [pk.PATH_INPUT, pk.PATH_AUDIO, pk.PATH_INPUT, pk.PATH_AUDIO] and set(paths[4:]) == {pk.PATH_BOTH}
    assert f32.ops[1].in0 == pk.SLOT_AUDIO and f32.ops[0].in0 == pk.SLOT_INPUT
    # runner boundary: mel -> front block; audio: STFT+mel -> front block that finalises the raw mel energies while loading
    assert f32.ops[2].in0 == f32.ops[0].out and f32.ops[3].in0 == f32.ops[1].out and f32.ops[2].out == f32.ops[3].out
    assert (f32.ops[2].p[9], f32.ops[3].p[9]) == (0, 1)
    assert sum(1 for o in f32.ops if o.kind == pk.F32_DWPW and o.p[12]) == 7  # residual blocks
    kinds = [pk.KIND_NAMES[o.kind] for o in i8.ops]
    assert kinds == ["i8_dwpw", "i8_front"] + ["i8_dwpw"] * 4 + ["i8_mid"] + ["i8_dwpw"] * 6 + ["i8_mean", "i8_fc", "i8_head", "i8_tail"]
    # the fused stage-2 chain covers stage2_ds1..ds3 (three operators, kept in the plan for the i8_mid = 0 path) and sits right behind them
    mid = i8.ops[6]
    assert mid.p[pk.TAIL_TAG] == pk.MID_OP and [o.p[pk.TAIL_TAG] == pk.MID_COVERED for o in i8.ops[:6]] == [False] * 3 + [True] * 3
    assert mid.in0 == i8.ops[3].in0 and mid.out == i8.ops[5].out and mid.p[5:11] == [3, 32, 64, 32, 512, 64]
    assert i8.tensors[mid.t[1]].size == 32 * 3
    # the fused tail operator covers stage 3-4 + MEAN + FC + head (9 operators, kept in the plan for the i8_tail = 0 path)
    tail = i8.ops[-1]
    assert tail.p[pk.TAIL_TAG] == pk.TAIL_OP and [o.p[pk.TAIL_TAG] == pk.TAIL_COVERED for o in i8.ops[:-1]] == [False] * 7 + [True] * 9
    assert tail.in0 == i8.ops[7].in0 and tail.out == pk.SLOT_SCORES and tail.p[:11] == [16 * 32 * 64, 10485760, 626688, 32 * 256 + 256 * 100, 100, 6, 16, 32, 64, 32, 256]
    desc = i8.tensors[tail.t[1]]
    assert desc.size == 24 * 6 + 16 and desc[:10].tolist() == [16, 32, 64, 128, 2, 8, 16, 0, 0, 0] and desc[24 * 5 : 24 * 5 + 10].tolist() == [4, 8, 256, 256, 1, 4, 8, 1, 1, 1]
    # pointwise A fragments of a tail block: lane (m, kq), byte b of k-step ks = W[16 nt + m][4 (base[kq] + 4 ks + (b >> 2)) + (b & 3)]
    from birdnet_stm32.models._lower_i8 import _tail_quad_base

    blk = i8.ops[8]  # stage3_ds2: 128 -> 128
    w2 = np.zeros((128, 128), np.int8)
    fr_old = i8.tensors[blk.t[4]].reshape(2, 8, 64, 16)  # generic fragment order [K/64][N/16][lane][16]: lane (q, c) -> W[16 ct + c][64 s + 16 q ..]
    for s_ in range(2):
        for ct in range(8):
            for lane in range(64):
                w2[16 * ct + (lane & 15), 64 * s_ + 16 * (lane >> 4) : 64 * s_ + 16 * (lane >> 4) + 16] = fr_old[s_, ct, lane]
    g_w = int(desc[24 * 1 + 20])
    frag = i8.tensors[tail.t[0]][g_w : g_w + 128 * 128 // 4].view(np.int8).reshape(8, 2, 64, 16)
    base = _tail_quad_base(128)
    for nt, ks, lane, b in ((0, 0, 0, 0), (3, 1, 37, 9), (7, 1, 63, 15), (5, 0, 18, 6)):
        assert frag[nt, ks, lane, b] == w2[16 * nt + (lane & 15), 4 * (base[lane >> 4] + 4 * ks + (b >> 2)) + (b & 3)]
    _, i8_dbg2 = _plans(keep_all=True, fuse=True)
    assert all(o.kind != pk.I8_TAIL for o in i8_dbg2.ops)  # keep_all plans keep every tensor visible: no fused tail
    mel = i8.ops[0]  # the mel mixer: QUANTIZE fused into its load (float32 spectrogram in), transposed output + PWL table
    assert mel.in0 == pk.SLOT_INPUT and mel.p[36] == 1 and mel.p[5] == 257 and mel.p[30] == 1 and mel.p[34] == 1 and mel.f[0] > 0
    _, i8_dbg = _plans(keep_all=True, fuse=True)  # debug plans keep QUANTIZE as its own operator (its tensor can be compared)
    assert [pk.KIND_NAMES[o.kind] for o in i8_dbg.ops][:2] == ["i8_quant", "i8_dwpw"] and i8_dbg.ops[1].p[36] == 0
    assert sum(1 for o in i8.ops if o.kind == pk.I8_DWPW and o.p[18]) == 7
    for plan in (f32, i8):
        for o in plan.ops:
            if o.kind in (pk.F32_DWPW, pk.I8_DWPW):
                th, tw, nb = (o.p[16], o.p[17], o.p[18]) if o.kind == pk.F32_DWPW else (o.p[31], o.p[32], o.p[33])
                assert th * tw * nb == 64 and o.p[6] % th == 0 and o.p[7] % tw == 0
    # fragment-ordered pointwise weights: [K/16][N/16][64][4] floats hold exactly the folded [K][N] matrix
    from birdnet_stm32.models._lower_f32 import pack_pw_fragments
    from birdnet_stm32.models._lower_i8 import pack_i8_fragments

    w = np.arange(32 * 48, dtype=np.float32).reshape(32, 48)
    fr = pack_pw_fragments(w)
    assert fr.shape == (2, 3, 64, 4)
    for j, ct, lane, e in ((0, 0, 0, 0), (1, 2, 37, 3), (0, 1, 63, 2)):
        assert fr[j, ct, lane, e] == w[16 * j + 4 * (lane >> 4) + e, 16 * ct + (lane & 15)]
    w8 = (np.arange(32 * 80) % 251 - 125).astype(np.int8).reshape(32, 80)
    f8 = pack_i8_fragments(w8)
    assert f8.shape == (2, 2, 64, 16)
    assert np.array_equal(f8[1, 0, 5], w8[5, 64:80]) and np.all(f8[1, 1, 17] == 0)  # step 1: lane 5 -> k 64..79; lane 17 -> k 80..95 = padding
    assert np.array_equal(f8[0, 1, 37], w8[16 + 5, 32:48])  # lane 37: q = 2, c = 5 -> channel 21, k = 32..47
    assert np.all(f8[1, :, 16:, :] == 0)  # k >= 80 is zero padding (q >= 1 of step 1)


def test_plan_structure():
    from birdnet_stm32.models import _pack as pk

    f32, i8 = _plans()
    kinds = [pk.KIND_NAMES[o.kind] for o in f32.ops]
    assert kinds[:2] == ["f32_mel", "f32_stem"] and kinds[-2:] == ["f32_gap", "f32_dense"]
    assert kinds.count("f32_dw") == 11 and kinds.count("f32_pw") == 11 and len(f32.ops) == 26
    assert sum(1 for o in f32.ops if o.kind == pk.F32_PW and o.p[4]) == 7  # residual blocks
    kinds = [pk.KIND_NAMES[o.kind] for o in i8.ops]
    assert kinds[:3] == ["i8_quant", "i8_mel", "i8_stem"] and kinds[-3:] == ["i8_mean", "i8_fc", "i8_head"]
    assert kinds.count("i8_dw") == 11 and kinds.count("i8_pw") == 11 and len(i8.ops) == 28  # 56 TFLite ops -> 28 launches
    assert sum(1 for o in i8.ops if o.kind == pk.I8_PW and o.p[6]) == 7
    assert (f32.dtype, i8.dtype) == (pk.DTYPE_F32, pk.DTYPE_I8)
    assert f32.input_elems == i8.input_elems == 257 * 256 and f32.num_classes == i8.num_classes == 100
    # shapes of the SURVEY §8d per-layer table
    shapes = [o.out_shape for o in f32.ops if o.kind in (pk.F32_STEM, pk.F32_PW)]
    assert shapes[0] == (64, 128, 16) and shapes[1] == (32, 64, 32) and shapes[-1] == (4, 8, 256)
    # TensorFlow SAME padding is asymmetric for stride 2: stem pads W 0/1, stride-2 depthwise pads 0/1 on both axes
    stem = f32.ops[1]
    assert (stem.p[8], stem.p[9]) == (1, 0)
    dws = [o for o in f32.ops if o.kind == pk.F32_DW]
    assert all((o.p[8], o.p[9]) == ((0, 0) if o.p[3] == 2 else (1, 1)) for o in dws)


def test_slots_never_alias_live_values():
    """Liveness check of the packer: an operator's output slot differs from every slot still to be read."""
    for f32, i8 in (_plans(), _plans(fuse=True)):
      for plan in (f32, i8):
        assert len(plan.slot_bytes) <= 4
        for i, o in enumerate(plan.ops):
            if o.out < 0:
                continue
            live_inputs = {o.in0, o.in1} - {-9, -1, -4}
            assert o.out not in live_inputs or o.kind == 2, f"op {i} writes its own input slot"
    keep, _ = _plans(keep_all=True)
    outs = [o.out for o in keep.ops if o.out >= 0]
    assert len(outs) == len(set(outs))  # debug plans keep every activation


def test_front_block_pair_is_tagged_and_never_runs_in_place():
    """The front block and the residual block behind it may run as one kernel (``f32_front2_kernel``): the packer tags the pair only in
    production plans, and keeps the front block's INPUT slots apart from the pair's output slot — the fused kernel reads the input of
    chunk 4 b .. 4 b + 3 while another workgroup already writes the (four times larger) output of chunk b."""
    from birdnet_stm32.models import _pack as pk

    plan = _plans(fuse=True)[0]
    heads = [(i, o) for i, o in enumerate(plan.ops) if o.p[pk.TAIL_TAG] == pk.FRONT2_HEAD]
    covered = [(i, o) for i, o in enumerate(plan.ops) if o.p[pk.TAIL_TAG] == pk.FRONT2_COVERED]
    assert len(heads) == 2 and len(covered) == 1  # spectrogram entry and audio entry share the residual block
    ci, c = covered[0]
    assert c.kind == pk.F32_DWPW and c.in0 == c.in1
    for hi, h in heads:
        assert h.kind == pk.F32_FRONT and hi + h.p[pk.FRONT2_DIST] == ci and h.out == c.in0
        assert c.out != h.in0 and c.out != h.out
    # nobody else reads the map between them
    assert not [i for i, o in enumerate(plan.ops) if heads[0][0] < i < ci and c.in0 in (o.in0, o.in1)]
    for other in (_plans()[0], _plans(keep_all=True)[0]):  # unfused lowering, debug plan: no tags
        assert not [o for o in other.ops if o.p[pk.TAIL_TAG] in (pk.FRONT2_HEAD, pk.FRONT2_COVERED)]


def test_residual_source_survives_until_the_add():
    from birdnet_stm32.models import _pack as pk

    for f32 in (_plans()[0], _plans(fuse=True)[0]):
      for i, o in enumerate(f32.ops):
        if (o.kind == pk.F32_PW and o.p[4]) or (o.kind == pk.F32_DWPW and o.p[12]):
            res = o.in1
            # the producer of `res` is the last writer of that slot before op i
            writers = [j for j in range(i) if f32.ops[j].out == res]
            assert writers, "residual slot has no producer"
            last = writers[-1]
            assert all(f32.ops[j].out != res for j in range(last + 1, i)), "residual overwritten before its add"


def test_blob_layout_roundtrip():
    from birdnet_stm32.models import _pack as pk

    _, i8 = _plans()
    blob = i8.to_blob()
    magic, ver, dtype, kind, elems, F, W, C, n_slots, n_tensors, n_ops, so, to, oo, _ = struct.unpack_from("<8s14I", blob, 0)
    assert magic == pk.BLOB_MAGIC and ver == pk.BLOB_VERSION and (dtype, kind, elems, F, W, C) == (1, 0, 257 * 256, 257, 256, 100)
    assert (n_slots, n_tensors, n_ops) == (len(i8.slot_bytes), len(i8.tensors), len(i8.ops))
    op_size = 16 + 4 * (pk.OP_NP + pk.OP_NT + pk.OP_NF)
    for t in range(n_tensors):
        off, nb = struct.unpack_from("<QQ", blob, to + 16 * t)
        assert off % 256 == 0 and off + nb <= len(blob)
        assert blob[off : off + nb] == i8.tensors[t].tobytes()
    k0 = struct.unpack_from("<4i", blob, oo)
    assert k0[0] == pk.I8_QUANT and k0[1] == pk.SLOT_INPUT
    assert oo + op_size * n_ops <= min(struct.unpack_from("<Q", blob, to)[0], len(blob))
    # the header constants mirror csrc/bn_blob.h
    hdr = open(os.path.join(REPO, "birdnet-stm32_amd", "csrc", "bn_blob.h")).read()
    assert f'"{pk.BLOB_MAGIC.decode()}"' in hdr and f"BN_BLOB_VERSION {pk.BLOB_VERSION}u" in hdr
    assert f"BN_OP_NP {pk.OP_NP}" in hdr and f"BN_OP_NT {pk.OP_NT}" in hdr and f"BN_OP_NF {pk.OP_NF}" in hdr
    for name, val in pk.KIND_NAMES.items():
        assert re.search(rf"BN_OP_{val.upper()} = {name}\b", hdr), val


def test_band_sparse_mel_equals_dense():
    from birdnet_stm32.models._keras_loader import load_keras_archive
    from birdnet_stm32.models._lower_f32 import mel_bands

    mel = load_keras_archive(KERAS_PATH).frontend.weights["mel"]
    vals, bands = mel_bands(mel, 257)
    dense = np.zeros((257, 64), np.float32)
    for m in range(64):
        s, n, off = bands[:, m]
        dense[s : s + n, m] = vals[off : off + n]
    assert np.array_equal(dense, mel[:257])
    assert vals.size < 0.05 * mel.size  # the Slaney triangles are narrow
    full = np.random.default_rng(0).uniform(0.1, 1, (264, 8)).astype(np.float32)
    v2, b2 = mel_bands(full, 257)
    assert np.all(b2[1] == 257) and v2.size == 257 * 8  # a dense mixer degrades to full-length bands


def test_pwl_table_and_folded_biases_match_the_oracle():
    """The lowering's own integer arithmetic (tables, multipliers, zero-point folding) against the oracle's."""
    from birdnet_stm32.models import _pack as pk
    from birdnet_stm32.models import _quant as qz
    from birdnet_stm32.models._tflite_reader import load_tflite
    from oracle import int8_graph as og

    model = load_tflite(TFLITE_PATH)
    _, i8f = _plans(fuse=True)
    _, i8 = _plans()
    assert np.array_equal(i8f.tensors[i8f.ops[0].t[8]], i8.tensors[i8.ops[1].t[4]])  # same PWL table in both plans (fused: mel mixer first)
    mel = i8.ops[1]
    lut = i8.tensors[mel.t[4]]  # [64][256]
    interp = og.Int8Interpreter(model)
    q = np.repeat(np.arange(-128, 128, dtype=np.int8)[None, None, :, None], 64, axis=3)  # [1,1,256,64]: every value in every channel
    env = {83: q}
    for op in model.ops[9:20]:
        env[op.outputs[0]] = interp._conv(op, env, True) if op.name == "DEPTHWISE_CONV_2D" else interp._add(op, env)
    assert np.array_equal(env[94][0, 0].T, lut)
    rng = np.random.default_rng(1)
    for real in list(rng.uniform(1e-6, 0.9, 50)) + [0.5, 0.25, 1e-9, 0.999999999]:
        assert qz.quantize_multiplier(float(real)) == og.quantize_multiplier(float(real))
    acc = rng.integers(-(2**26), 2**26, 4096)
    for m, s in ((1518500250, -7), (1073741824, 0), (2147483647, -1), (1385918850, -2), (1200000000, 1)):
        assert np.array_equal(qz.requantize(acc, m, s), og.mbqm(acc, m, s))
    for act in ("none", "relu", "relu6"):
        for sc, zp in ((0.0235294, -128), (0.20542, -34), (1.23353e-3, -128)):
            assert qz.activation_bounds(act, sc, zp) == og.activation_range(act, sc, zp)
    head = i8.ops[-1]
    assert np.array_equal(i8.tensors[head.t[0]], interp.logistic_lut(model.ops[54]))
    # zero-point folding: bias' = bias - zp_in * sum_k w
    pw = next(o for o in i8.ops if o.kind == pk.I8_PW)
    op = model.ops[24]
    w = model.tensors[op.inputs[1]].data.reshape(32, 16).astype(np.int64)
    zp_in = int(model.tensors[op.inputs[0]].zero_point[0])
    assert np.array_equal(i8.tensors[pw.t[1]], (model.tensors[op.inputs[2]].data - zp_in * w.sum(axis=1)).astype(np.int32))


def test_unsupported_graphs_are_rejected():
    from birdnet_stm32.models import build_model
    from birdnet_stm32.models._h5_reader import H5File
    from birdnet_stm32.models._lower_f32 import lower_f32
    from birdnet_stm32.models._tflite_reader import parse_tflite

    with pytest.raises(ValueError, match="TFL3"):
        parse_tflite(b"\x00" * 64)
    with pytest.raises(ValueError, match="HDF5"):
        H5File(b"\x00" * 128)
    kw = dict(num_mels=64, spec_width=256, sample_rate=22050, chunk_duration=3, embeddings_size=256, num_classes=10)
    # precomputed frontends lower to a plan that starts at the stem (the graph passes the host-side map through)
    plan = lower_f32(build_model("dscnn", audio_frontend="librosa", use_se=False, use_inverted_residual=False, **kw))
    assert plan.input_kind == 2 and plan.input_elems == 64 * 256 and plan.ops[0].in0 == -1
    plan = lower_f32(build_model("dscnn", audio_frontend="mfcc", n_mfcc=20, **kw))
    assert plan.input_kind == 2 and plan.input_elems == 20 * 256
    spec = build_model("dscnn", audio_frontend="librosa", **kw)
    spec.frontend.attrs["mode"] = "cqt"
    with pytest.raises(NotImplementedError):
        lower_f32(spec)


# ------------------------------------------------------------------------------------------ C ABI
def _declared_symbols():
    hdr = open(os.path.join(REPO, "include", "birdnet_hip.h")).read()
    return sorted(set(re.findall(r"BN_API [\w\s\*]+?(bn_\w+)\(", hdr)))


def test_library_exports_every_declared_symbol():
    from birdnet_stm32 import _hip

    assert os.path.isfile(_hip.LIB_PATH), "libbirdnet_hip.so missing: run __graft_entry__.build()"
    declared = _declared_symbols()
    assert len(declared) >= 15 and sorted(_hip.EXPORTS) == declared
    lib = ctypes.CDLL(_hip.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    lib.bn_version.restype = ctypes.c_int
    assert lib.bn_version() == _hip.ABI_VERSION
    lib.bn_kernel_names.restype = ctypes.c_char_p
    names = lib.bn_kernel_names().decode().split("\n")
    assert "stft512_mag_kernel" in names and "i8_pw_kernel" in names


def test_launcher_options_round_trip_without_a_device():
    """bn_set_option / bn_get_option are plain host state (no compute): every documented name exists with its production
    default, values round-trip, unknown names are refused, and no launcher reads the environment any more."""
    from birdnet_stm32 import _hip

    defaults = {"f32_strip": 1, "f32_strip_th": 0, "f32_front_staged": 1, "f32_front2": 1, "f32_pwdw": 2, "f32_tile_slice": 0, "f32_pw_ws": 1, "i8_pwdw": 0, "i8_pw_lds": 1, "i8_pw_forms": 1, "i8_add_tab": 1, "front_tpw": 0, "wave_dwpw": 1, "i8_strip": 1, "i8_strip_mfdw": 1, "i8_strip_th": 0, "i8_dw_pool": 1, "i8_tail_fclds": 1,
                "i8_tail": 1, "i8_tail_mfdw": 1, "i8_mid": 1, "i8_mel_generic": 0, "stft_rowmajor": 0, "stft_exact": 2, "stft_flagcap": 1022, "stft_guard": 0, "stft_audit": 0, "stft_minint": 1, "ingest_blk": 0, "ingest_generic": 0}
    assert sorted(defaults) == sorted(_hip.OPTION_NAMES)
    hdr = open(os.path.join(REPO, "include", "birdnet_hip.h")).read()
    for name, want in defaults.items():
        assert f'"{name}"' in hdr, name
        if not os.environ.get("BN_" + name.upper()):
            assert _hip.get_option(name) == want, name
        with _hip.options(**{name: 7}):
            assert _hip.get_option(name) == 7
        assert _hip.get_option(name) == (want if not os.environ.get("BN_" + name.upper()) else int(os.environ["BN_" + name.upper()]))
    with pytest.raises(_hip.HipError, match="unknown option"):
        _hip.set_option("no_such_switch", 1)
    csrc = os.path.join(PKG, "csrc")
    for fn in os.listdir(csrc):
        if fn.endswith(".hip") and fn != "bn_api.hip":
            assert "getenv" not in open(os.path.join(csrc, fn)).read(), f"{fn} reads the environment on a launch path"
    assert open(os.path.join(csrc, "bn_api.hip")).read().count("getenv(") == 1  # options_from_env, once at load


def _all_plans():
    from birdnet_stm32.models import build_model
    from birdnet_stm32.models._lower_f32 import lower_f32
    from birdnet_stm32.models.runners import lower_model_file

    for path in (KERAS_PATH, TFLITE_PATH):
        for keep_all in (False, True):
            for fuse in (True, False):
                yield f"{os.path.basename(path)} keep_all={keep_all} fuse={fuse}", lower_model_file(path, keep_all=keep_all, fuse=fuse)
    base = dict(num_mels=64, spec_width=256, sample_rate=24000, chunk_duration=3, embeddings_size=256, num_classes=10, randomize_bn=True, seed=7)
    for kw in (dict(), dict(use_inverted_residual=False, use_se=True, embeddings_size=128, use_attention_pooling=True, class_activation="sigmoid"),
               dict(alpha=1.5, mag_scale="pcen", num_classes=37), dict(use_se=False, depth_multiplier=2, alpha=0.5, mag_scale="none"),
               dict(num_mels=32, spec_width=128, sample_rate=16000, chunk_duration=2), dict(num_mels=48, spec_width=192, sample_rate=22050, alpha=0.75),
               dict(chunk_duration=2, audio_frontend="raw", mag_scale="pcen", alpha=1.5), dict(audio_frontend="librosa"), dict(audio_frontend="mfcc", num_mels=20)):
        spec = build_model("dscnn", **{**base, **kw})
        for fuse in (True, False):
            yield f"dscnn {kw} fuse={fuse}", lower_f32(spec, fuse=fuse)


def test_inverted_residual_pairs_are_tagged_and_never_run_in_place():
    """Expand 1x1 + depthwise 3x3 of inverted-residual blocks may run as one kernel (``f32_pwdw_kernel``): the packer tags the pair only in
    production plans, only when the depthwise stage is the sole reader of the expanded map, and keeps the block input's slot away from
    the depthwise output (the fused kernel reads one while it writes the other)."""
    from birdnet_stm32.models import _pack as pk
    from birdnet_stm32.models import build_model
    from birdnet_stm32.models._lower_f32 import lower_f32

    spec = build_model("dscnn", num_mels=64, spec_width=256, sample_rate=24000, chunk_duration=2, embeddings_size=256, num_classes=10,
                       audio_frontend="raw", mag_scale="pcen", alpha=1.5, use_se=True, use_inverted_residual=True)
    plan = lower_f32(spec)
    heads = [i for i, o in enumerate(plan.ops) if o.p[pk.TAIL_TAG] == pk.PWDW_HEAD]
    assert len(heads) == 11
    for i in heads:
        e, d = plan.ops[i], plan.ops[i + 1]
        assert e.kind == pk.F32_DWPW and e.p[15] == 0 and e.p[12] == 0 and e.p[13] == 0 and d.kind == pk.F32_DW and d.p[pk.TAIL_TAG] == pk.PWDW_COVERED
        assert d.in0 == e.out and d.out != e.in0 and d.out != e.out and d.p[2] == e.p[10]
    stems = [i for i, o in enumerate(plan.ops) if o.p[pk.TAIL_TAG] == pk.PWDW_STEM]  # the stem in front of the first pair: computed inside the fused kernel
    assert len(stems) == 1 and plan.ops[stems[0]].kind == pk.F32_STEM and stems[0] + 1 == heads[0] and plan.ops[stems[0] + 2].out != plan.ops[stems[0]].in0
    for kw in (dict(keep_all=True), dict(fuse=False)):
        assert not any(o.p[pk.TAIL_TAG] in (pk.PWDW_HEAD, pk.PWDW_COVERED, pk.PWDW_STEM) for o in lower_f32(spec, **kw).ops)


def test_blob_check_accepts_every_lowered_plan_and_refuses_damaged_ones():
    """bn_blob_check (host only): the validation bn_model_load runs before it touches the device.  Every plan the lowering passes
    produce passes; a slot made too small for an operator's output, a gate / residual slot id outside the plan, a slot without storage,
    a truncated constant tensor, an impossible geometry and a truncated blob are all refused with BN_ERR_FORMAT."""
    from birdnet_stm32 import _hip
    from birdnet_stm32.models import _pack as pk

    plans = dict(_all_plans())
    assert len(plans) >= 20
    for name, plan in plans.items():
        try:
            _hip.blob_check(plan.to_blob())
        except _hip.HipError as e:  # pragma: no cover
            raise AssertionError(f"{name}: {e}") from e

    import copy

    def refused(plan, why):
        with pytest.raises(_hip.HipError, match=why):
            _hip.blob_check(plan.to_blob())

    i8 = plans[f"{os.path.basename(TFLITE_PATH)} keep_all=True fuse=True"]
    f32 = next(p for n, p in plans.items() if n.startswith("dscnn {} fuse=True"))
    # 1) an output slot smaller than what the operator writes
    bad = copy.deepcopy(i8)
    oi = next(i for i, o in enumerate(bad.ops) if o.kind == pk.I8_DWPW and o.p[29])
    bad.slot_bytes[bad.ops[oi].out] = 256
    refused(bad, "needs .* bytes per chunk")
    # 2) a gate slot id stored in p[] that is not a slot of the plan (squeeze-excite)
    bad = copy.deepcopy(f32)
    gated = [i for i, o in enumerate(bad.ops) if (o.kind == pk.F32_DWPW and o.p[13]) or (o.kind == pk.F32_PW and o.p[5])]
    assert gated
    o = bad.ops[gated[0]]
    o.p[14 if o.kind == pk.F32_DWPW else 6] = 10_000
    refused(bad, "slot id 10000 is not a slot")
    # 3) a referenced slot without storage
    bad = copy.deepcopy(i8)
    bad.slot_bytes[bad.ops[2].in0] = 0
    refused(bad, "has no storage")
    # 4) a constant tensor shorter than the operator reads
    bad = copy.deepcopy(i8)
    o = bad.ops[oi]
    bad.tensors[o.t[4]] = bad.tensors[o.t[4]].reshape(-1)[:64].copy()
    refused(bad, "holds 64 bytes, the operator reads")
    # 5) geometry that does not follow from the input size
    bad = copy.deepcopy(i8)
    bad.ops[oi].p[6] -= 1
    refused(bad, "does not follow from input")
    # 6) residual slot missing
    bad = copy.deepcopy(i8)
    oj = next(i for i, o in enumerate(bad.ops) if o.kind == pk.I8_DWPW and o.p[18])
    bad.ops[oj].in1 = pk.SLOT_NONE
    refused(bad, "residual")
    # 6b) flags that select a kernel form the block's other fields do not describe
    bad = copy.deepcopy(i8)
    bad.ops[oi].p[30] = 1  # transposed output on a depthwise + pointwise block
    refused(bad, "not a mel mixer")
    bad = copy.deepcopy(i8)
    bad.ops[oi].p[34] = 1  # per-channel table on a block whose kernel has none
    refused(bad, "table")
    # 6c) a clamp that is not an int8 range on a value that indexes a 256-entry table (residual ADD tables, the mel mixer's PWL table)
    bad = copy.deepcopy(i8)
    bad.ops[oj].p[28] = 300
    refused(bad, "not an int8 range")
    bad = copy.deepcopy(i8)
    om = next(i for i, o in enumerate(bad.ops) if o.kind == pk.I8_DWPW and o.p[34])
    bad.ops[om].p[16] = -129
    refused(bad, "not an int8 range")
    # 7) truncated blob, bad magic
    blob = i8.to_blob()
    for cut in (10, 63, 200, len(blob) // 2):
        assert _hip.load_library().bn_blob_check(blob[:cut], cut) == -3
    assert _hip.load_library().bn_blob_check(b"XXXXXXXX" + blob[8:], len(blob)) == -3


def test_context_creation_fails_loudly_without_device():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from birdnet_stm32 import _hip

    lib = _hip.load_library()
    assert lib.bn_device_count() == 0
    with pytest.raises(_hip.HipError, match="no HIP device|no CPU fallback"):
        _hip.Context(0, 4)
    h = ctypes.c_void_p()
    assert lib.bn_ctx_create(0, 0, ctypes.byref(h)) < 0 and b"max_batch" in lib.bn_last_error()


# ------------------------------------------------------------------ strip-kernel constant block (bn_i8_strip.hip)
def _emulate_strip(cst, x, p, nw, tab=None):
    """numpy restatement of ``i8_strip_kernel``'s lane arithmetic (csrc/bn_i8_strip.hip) from the packer's constant block:
    wave w / lane (n, kq) owns channels CW w + CL kq .. of column n, row-transposed depthwise weights, folded requantisation
    addends, A fragments with permuted rows per channel slice, own value + 128 as the table index of the ADD.
    x: int8 [B][H][W][C]."""
    H, W, C, S, OH, OW, pt, pl = p[0], p[1], p[2], p[3], p[6], p[7], p[8], p[9]
    z_in, dw_lo, dw_hi, N, pw_zp, pw_lo, pw_hi = p[10], p[12], p[13], p[14], p[15], p[16], p[17]
    add = p[18:29]
    CW, CWO = C // nw, N // nw
    CL, COL = CW // 4, CWO // 4
    QL, NT = CL // 4, CWO // 16
    sizes = [nw * 4 * QL * 12, nw * 4 * QL * 4, nw * 4 * QL * 12, nw * NT * nw * 64 * QL, nw * 4 * NT * 4, nw * 4 * NT * 12]
    assert cst.size == sum(sizes)
    o = np.concatenate([[0], np.cumsum(sizes)])
    dww = cst[o[0]:o[1]].view(np.int8).reshape(nw, 4, QL, 3, 4, 4).astype(np.int64)  # [w][kq][ql][row][e][tap byte]
    dwb = cst[o[1]:o[2]].reshape(nw, 4, QL, 4).astype(np.int64)
    dwc = cst[o[2]:o[3]].reshape(nw, 4, QL, 3, 4).astype(np.int64)
    pwa = cst[o[3]:o[4]].view(np.int8).reshape(nw, NT, nw, 64, CL).astype(np.int64)
    pwb = cst[o[4]:o[5]].reshape(nw, 4, NT, 4).astype(np.int64)
    pwc = cst[o[5]:o[6]].reshape(nw, 4, NT, 3, 4).astype(np.int64)
    assert np.all(dww[..., 3] == 0)

    def rq(v, m, c1, e):
        t = (v * m + (1 << 30)) >> 31
        r = t + c1 + (t >> 63)
        assert np.abs(r).max() < 2**31 and np.abs(t).max() < 2**31
        return r >> e

    B = x.shape[0]
    xp = np.full((B, H + 2 + S, W + 2 + S, C), z_in, np.int64)
    xp[:, pt:pt + H, pl:pl + W] = x
    bfrag = np.zeros((B, OH, OW, nw, 4, CL), np.int64)  # [.. position][wave][kq][k byte of the lane]
    for w in range(nw):
        for kq in range(4):
            for ql in range(QL):
                for e in range(4):
                    c = CW * w + CL * kq + 4 * ql + e
                    acc = np.full((B, OH, OW), dwb[w, kq, ql, e])
                    for i in range(3):
                        for j in range(3):
                            acc = acc + xp[:, i:i + S * OH:S, j:j + S * OW:S, c] * dww[w, kq, ql, i, e, j]
                    m, c1, sh = dwc[w, kq, ql, :, e]
                    bfrag[..., w, kq, 4 * ql + e] = np.clip(rq(acc, m, c1, sh), dw_lo, dw_hi)
    y = np.zeros((B, OH, OW, N), np.int64)
    off = 128 if add[0] else 0
    for w in range(nw):
        for t in range(NT):
            for mrow in range(16):
                q, reg = mrow >> 2, mrow & 3
                acc = np.full((B, OH, OW), pwb[w, q, t, reg])
                for ks in range(nw):
                    for kq in range(4):
                        acc = acc + (bfrag[..., ks, kq, :] * pwa[w, t, ks, 16 * kq + mrow]).sum(axis=-1)
                m, c1, sh = pwc[w, q, t, :, reg]
                v = np.clip(rq(acc, m, c1, sh), pw_lo + off, pw_hi + off)
                ch = CWO * w + COL * q + 4 * t + reg
                if add[0]:
                    res = x[..., ch].view(np.uint8).astype(np.int64)  # centre tap of the lane's own channel group
                    v = tab[res, v]  # the ADD as a two-byte table
                y[..., ch] = v
    return y.astype(np.int8)


def test_strip_constant_block_reproduces_the_oracle():
    """Every block the packer marks for the strip kernel (wide early layers of the shipped INT8 graph): the kernel's
    arithmetic, restated in numpy from the constant block, gives the oracle's tensors bit for bit — dead channels
    (shifts of 26-31 bits, rewritten as constants) and the + 128 table-index form included."""
    from birdnet_stm32.models import _pack as pk
    from birdnet_stm32.models._tflite_reader import load_tflite
    from birdnet_stm32.models.runners import lower_model_file
    from oracle import stft
    from oracle.int8_graph import Int8Interpreter

    from conftest import synth_chunks

    plan = lower_model_file(TFLITE_PATH, keep_all=True, fuse=True)
    strip_ops = [o for o in plan.ops if o.kind == pk.I8_DWPW and o.p[35]]
    assert [o.name for o in strip_ops] == ["t102", "t104", "t107", "t110", "t112", "t115", "t118", "t121", "t123", "t126"]
    S = np.stack([stft.hybrid_spectrogram(a) for a in synth_chunks(2)])[..., None]
    _, env = Int8Interpreter(load_tflite(TFLITE_PATH)).invoke(S, return_all=True)
    by_val = {o.out: o for o in plan.ops}
    for o in strip_ops:
        src = by_val[o.in0]
        x = env[int(src.name[1:])].reshape(2, o.p[0], o.p[1], o.p[2])
        want = env[int(o.name[1:])].reshape(2, o.p[6], o.p[7], o.p[14])
        if o.p[18]:
            assert o.in1 == o.in0  # the residual is the block input
        from birdnet_stm32.models._lower_i8 import strip_waves

        nw = strip_waves(o.p[2], o.p[14], o.p[3], o.p[7], bool(o.p[18]))
        assert nw == {32: 1, 64: 2, 128: 4, 256: 8}[o.p[2]]
        tab = np.asarray(plan.tensors[o.t[10]], np.int8).reshape(256, 256) if o.p[18] else None
        got = _emulate_strip(np.asarray(plan.tensors[o.t[9]], np.int32), x, o.p, nw, tab)
        assert np.array_equal(got, want), f"{o.name}: {(got != want).sum()} of {got.size} values differ"


def _emulate_front_strip(cst, fe, p):
    """numpy restatement of ``i8_front_strip_kernel`` from its constant block: the stem as a matrix product with contraction
    index 8 * (window row) + column, lane (n, kq) reading input row (stem row - 1 + kq).  fe: int8 [B][H0][W0]."""
    H0, W0, OH, OW = p[0], p[1], p[4], p[5]
    z_fe, z_st, st_lo, st_hi, dw_lo, dw_hi, pw_lo, pw_hi = p[6], p[7], p[8], p[9], p[11], p[12], p[14], p[15]
    sta = cst[0:64].view(np.int8).reshape(64, 4).astype(np.int64)
    stb = cst[64:80].reshape(4, 4).astype(np.int64)
    stc = cst[80:128].reshape(4, 3, 4).astype(np.int64)
    dww = cst[128:176].view(np.int8).reshape(4, 3, 4, 4).astype(np.int64)
    dwb = cst[176:192].reshape(4, 4).astype(np.int64)
    dwc = cst[192:240].reshape(4, 3, 4).astype(np.int64)
    pwa = cst[240:368].view(np.int8).reshape(2, 64, 4).astype(np.int64)
    pwb = cst[368:400].reshape(4, 2, 4).astype(np.int64)
    pwc = cst[400:496].reshape(4, 2, 3, 4).astype(np.int64)
    assert np.all(sta[48:] == 0) and np.all(sta[:, 3] == 0)

    def rq(v, m, c1, e):
        t = (v * m + (1 << 30)) >> 31
        r = t + c1 + (t >> 63)
        assert np.abs(r).max() < 2**31 and np.abs(t).max() < 2**31
        return r >> e

    B = fe.shape[0]
    fp = np.full((B, H0 + 2, W0 + 2), z_fe, np.int64)
    fp[:, 1:H0 + 1, :W0] = fe
    SH, SW = H0, W0 // 2
    stem = np.full((B, SH + 1, SW + 1, 16), z_st, np.int64)
    for q in range(4):
        for reg in range(4):
            c = 4 * q + reg
            acc = np.full((B, SH, SW), stb[q, reg])
            for kq in range(3):
                for j in range(3):
                    acc = acc + fp[:, kq:kq + SH, j:j + 2 * SW:2] * sta[16 * kq + c, j]
            stem[:, :SH, :SW, c] = np.clip(rq(acc, *stc[q, :, reg]), st_lo, st_hi)
    dwq = np.zeros((B, OH, OW, 16), np.int64)
    for kq in range(4):
        for e in range(4):
            acc = np.full((B, OH, OW), dwb[kq, e])
            for i in range(3):
                for j in range(3):
                    acc = acc + stem[:, i:i + 2 * OH:2, j:j + 2 * OW:2, 4 * kq + e] * dww[kq, i, e, j]
            dwq[..., 4 * kq + e] = np.clip(rq(acc, *dwc[kq, :, e]), dw_lo, dw_hi)
    y = np.zeros((B, OH, OW, 32), np.int64)
    for q in range(4):
        for t in range(2):
            for reg in range(4):
                acc = np.full((B, OH, OW), pwb[q, t, reg])
                for kq in range(4):
                    acc = acc + (dwq[..., 4 * kq:4 * kq + 4] * pwa[t, 16 * kq + 4 * q + reg]).sum(axis=-1)
                y[..., 8 * q + 4 * t + reg] = np.clip(rq(acc, *pwc[q, t, :, reg]), pw_lo, pw_hi)
    return y.astype(np.int8)


def test_add_table_is_the_oracles_add_on_every_byte_pair():
    """``_lower_i8.add_table`` (the 64 KB table the strip / pointwise kernels look the residual ADD up in) against the oracle's int8 ADD (TFLite's
    left-shift-20 form, oracle/int8_graph.py) on all 65 536 (residual byte, own byte) pairs, for random scales / zero points / activations and
    for both operand orders; and the 1x1-convolution + ADD operators of an exported inverted-residual plan carry the table."""
    from birdnet_stm32.models import _pack as pk
    from birdnet_stm32.models import _quant as qz
    from birdnet_stm32.models._lower_i8 import add_table, lower_i8
    from oracle.int8_graph import activation_range, mbqm, quantize_multiplier

    rng = np.random.default_rng(11)
    res = np.arange(256).astype(np.uint8).view(np.int8).astype(np.int64)   # row index = the residual's BYTE PATTERN
    own = np.arange(256, dtype=np.int64) - 128                             # column index = own value + 128
    for trial in range(12):
        s1, s2, so = (float(np.float32(v)) for v in rng.uniform(0.004, 0.2, 3))
        z1, z2, zo = (int(v) for v in rng.integers(-128, 127, 3))
        act = ["none", "relu", "relu6"][trial % 3]
        twice_max = 2.0 * max(s1, s2)
        m1, m2, mo = quantize_multiplier(s1 / twice_max), quantize_multiplier(s2 / twice_max), quantize_multiplier(twice_max / ((1 << 20) * so))
        lo, hi = activation_range(act, so, zo)
        want = np.clip(mbqm((mbqm((res - z1) << 20, *m1))[:, None] + (mbqm((own - z2) << 20, *m2))[None, :], *mo) + zo, lo, hi).astype(np.int8)
        ap = qz.AddParams(s1, z1, s2, z2, so, zo, act)                     # residual = first operand
        got = add_table([1, ap.z1, ap.m1, ap.sh1, ap.m2, ap.sh2, ap.mo, ap.sho, ap.zo, ap.amin, ap.amax], z2)
        assert got.shape == (256, 256) and got.dtype == np.int8 and np.array_equal(got, want), trial
        ap = qz.AddParams(s2, z2, s1, z1, so, zo, act)                     # residual = second operand: the packer swaps the roles, not the arithmetic
        got = add_table([1, ap.z2, ap.m2, ap.sh2, ap.m1, ap.sh1, ap.mo, ap.sho, ap.zo, ap.amin, ap.amax], z2)
        assert np.array_equal(got, want), trial
    from test_conversion import EXPORT_TOPOLOGIES, _export

    _, model, _, _ = _export(next(v for k, v in EXPORT_TOPOLOGIES.items() if "ir" in k))
    plan = lower_i8(model)
    with_add = [o for o in plan.ops if o.kind == pk.I8_DWPW and o.p[18] and not o.p[29]]
    assert with_add and all(o.t[10] >= 0 and plan.tensors[o.t[10]].nbytes == 65536 for o in with_add)


def test_front_strip_constant_block_reproduces_the_oracle():
    """The INT8 front block (stem on the matrix cores + depthwise + pointwise) restated from the strip kernel's constant
    block: bit-identical to the oracle's tensor after the first pointwise convolution."""
    from birdnet_stm32.models import _pack as pk
    from birdnet_stm32.models._tflite_reader import load_tflite
    from birdnet_stm32.models.runners import lower_model_file
    from oracle import stft
    from oracle.int8_graph import Int8Interpreter

    from conftest import synth_chunks

    plan = lower_model_file(TFLITE_PATH, keep_all=True, fuse=True)
    (o,) = [o for o in plan.ops if o.kind == pk.I8_FRONT]
    assert o.p[16] == 1 and o.t[12] >= 0
    S = np.stack([stft.hybrid_spectrogram(a) for a in synth_chunks(2)])[..., None]
    _, env = Int8Interpreter(load_tflite(TFLITE_PATH)).invoke(S, return_all=True)
    src = {q.out: q for q in plan.ops}[o.in0]
    fe = env[int(src.name[1:])].reshape(2, o.p[0], o.p[1])
    want = env[int(o.name[1:])].reshape(2, o.p[4], o.p[5], o.p[3])
    got = _emulate_front_strip(np.asarray(plan.tensors[o.t[12]], np.int32), fe, o.p)
    assert np.array_equal(got, want), f"{(got != want).sum()} of {got.size} values differ"
