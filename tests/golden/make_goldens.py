"""Generate golden vectors by running the REAL reference (/root/reference) in this container.

    python tests/golden/make_goldens.py            # writes tests/golden/*.npz

The reference has no tests or fixtures of its own, so its outputs on seeded
synthetic weights/inputs are the only pin for the oracle.  Weights and inputs are NOT
stored: they are regenerated from seeds by rag-gesture_amd/synth.py (numpy Philox/PCG64,
platform independent); only expected outputs (and a few intermediate tensors) are stored.

All randomness of the reference is routed through synth.NoiseTape by patching
torch.randn / torch.randn_like / torch.distributions' _standard_normal for the duration
of each run, so the oracle and the HIP path can replay the identical noise
(SURVEY Appendix D gives the consumption order).

Runs only where /root/reference exists; nothing on the GPU box imports this file.
"""
import contextlib
import importlib
import os
import sys
import tempfile
import copy

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

import _ref_import  # noqa: E402

rg = importlib.import_module("rag-gesture_amd")
synth = rg.synth


@contextlib.contextmanager
def taped_noise(tape):
    """Route every normal draw of the reference through `tape` (SURVEY Appendix D order)."""
    import torch.distributions.normal as tdn
    import torch.distributions.utils as tdu

    o_randn, o_like, o_sn1, o_sn2 = torch.randn, torch.randn_like, tdn._standard_normal, tdu._standard_normal

    def randn(*shape, **kw):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        return tape.draw(shape)

    def randn_like(t, **kw):
        return tape.draw(tuple(t.shape))

    def std_normal(shape, dtype, device):
        return tape.draw(tuple(shape))

    torch.randn, torch.randn_like = randn, randn_like
    tdn._standard_normal = tdu._standard_normal = std_normal
    try:
        yield
    finally:
        torch.randn, torch.randn_like = o_randn, o_like
        tdn._standard_normal, tdu._standard_normal = o_sn1, o_sn2


def build_reference_model(ns, cfg, vae_cfgs, seed, tmpdir, inference_type="ddim"):
    """Instantiate the reference MotionDiffusion at `cfg` shapes with synthetic weights."""
    paths = {}
    for i, part in enumerate(synth.PARTS):
        d = os.path.join(tmpdir, part)
        os.makedirs(d, exist_ok=True)
        ypath = os.path.join(d, part + ".yaml")
        with open(ypath, "w") as f:
            yaml.safe_dump(vae_cfgs[part], f)
        sd = synth.synth_vae_state(seed + 101 + i, vae_cfgs[part])
        torch.save({"model_state": sd}, os.path.join(d, vae_cfgs[part]["test_ckpt"]))
        paths[part] = ypath
    d, te, H = cfg["latent_dim"], cfg["time_embed_dim"], cfg["num_heads"]
    model_cfg = dict(
        type="MotionDiffusion",
        model=dict(
            type="ReGestureTransformer", input_feats=189, max_seq_len=cfg["max_seq_len"],
            frame_chunk_size=cfg["frame_chunk_size"], latent_dim=d, time_embed_dim=te,
            num_layers=cfg["num_layers"], body_part_cat_axis="time",
            sa_block_cfg=dict(type="EfficientSelfAttention", latent_dim=d, num_heads=H, dropout=0, time_embed_dim=te),
            ca_block_cfg=dict(type="EfficientCrossAttention", latent_dim=d, text_latent_dim=d, num_heads=H,
                              dropout=0, time_embed_dim=te),
            ffn_cfg=dict(latent_dim=d, ffn_dim=cfg["ff_size"], dropout=0, time_embed_dim=te),
            vae_cfg=dict(upper_cfg=paths["upper"], lowertrans_cfg=paths["lowertrans"], face_cfg=paths["face"],
                         hands_cfg=paths["hands"], latent_dim=d, frame_chunk_size=cfg["frame_chunk_size"]),
            text_encoder=dict(pretrained_model=None, latent_dim=cfg["text_latent_dim"], num_layers=0,
                              ff_size=2048, dropout=0, use_text_proj=False),
            audio_encoder=dict(pretrained_model=None, latent_dim=cfg["text_latent_dim"], num_layers=0, dropout=0.1),
            speaker_embedding=dict(num_speakers=cfg["num_speakers"]),
            retrieval_train=False, retrieval_cfg=None,
            scale_func_cfg=dict(cfg["scale_func_cfg"]), per_joint_scale=dict(cfg["per_joint_scale"]),
        ),
        loss_recon=dict(type="MSELoss", loss_weight=1, reduction="none"),
        body_part_lossweights=dict(upper=1.0, hands=1.0, face=1.0, lowertransl=1.0),
        diffusion_train=dict(beta_scheduler="scaled_linear", diffusion_steps=1000, model_mean_type="start_x",
                             model_var_type="fixed_large"),
        diffusion_test=dict(beta_scheduler="scaled_linear", diffusion_steps=1000, model_mean_type="start_x",
                            model_var_type="fixed_large", respace="15,15,8,6,6", num_inference_timesteps=50,
                            classifier_free_guidance_scale=0),
        inference_type=inference_type,
    )
    model = ns.builder.build_architecture(model_cfg, database=None)
    full = synth.synth_full_state(seed, cfg, vae_cfgs)
    missing, unexpected = model.model.load_state_dict(full, strict=True), None
    model.eval()
    return model, full


def t2n(x):
    return x.detach().cpu().numpy()


def main():
    ns = _ref_import.load_reference()
    torch.set_num_threads(8)
    out = {}
    from oracle import denoiser as od, diffusion as odf, rotation as orot, vae as ovae

    # ---- (i) schedule tables -------------------------------------------------------------
    diff = ns.arch.build_diffusion(dict(beta_scheduler="scaled_linear", diffusion_steps=1000,
                                        model_mean_type="start_x", model_var_type="fixed_large",
                                        respace="15,15,8,6,6", num_inference_timesteps=50,
                                        classifier_free_guidance_scale=0))
    sch = odf.SpacedSchedule()
    assert sch.timestep_map == diff.timestep_map
    for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "alphas_cumprod_next",
              "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "sqrt_alphas_cumprod",
              "sqrt_one_minus_alphas_cumprod"):
        assert np.array_equal(getattr(sch, k), getattr(diff, k)), k
    np.savez(os.path.join(HERE, "schedule.npz"), timestep_map=np.array(diff.timestep_map),
             betas=diff.betas, alphas_cumprod=diff.alphas_cumprod,
             sqrt_recip_alphas_cumprod=diff.sqrt_recip_alphas_cumprod,
             sqrt_recipm1_alphas_cumprod=diff.sqrt_recipm1_alphas_cumprod)
    print("schedule ok")

    # ---- (ii) rotation conversions ---------------------------------------------------------
    g = np.random.Generator(np.random.PCG64(7))
    aa = g.uniform(-0.3, 0.3, size=(64, 3)).astype(np.float32)
    aa_small = (g.standard_normal((16, 3)) * 1e-8).astype(np.float32)
    axis = g.standard_normal((16, 3)); axis /= np.linalg.norm(axis, axis=-1, keepdims=True)
    aa_pi = (axis * (np.pi - g.uniform(0, 1e-3, size=(16, 1)))).astype(np.float32)
    aa_big = g.uniform(-2.5, 2.5, size=(32, 3)).astype(np.float32)
    aa_all = torch.from_numpy(np.concatenate([aa, aa_small, aa_pi, aa_big], 0))
    d6 = ns.rc.matrix_to_rotation_6d(ns.rc.axis_angle_to_matrix(aa_all))
    d6_in = torch.from_numpy(g.standard_normal((128, 6)).astype(np.float32))
    aa_out = ns.rc.matrix_to_axis_angle(ns.rc.rotation_6d_to_matrix(d6_in))
    aa_rt = ns.rc.matrix_to_axis_angle(ns.rc.rotation_6d_to_matrix(d6))
    np.savez(os.path.join(HERE, "rotation.npz"), aa_in=t2n(aa_all), d6_out=t2n(d6), d6_in=t2n(d6_in),
             aa_out=t2n(aa_out), aa_roundtrip=t2n(aa_rt))
    e1 = (orot.matrix_to_rotation_6d(orot.axis_angle_to_matrix(aa_all)) - d6).abs().max().item()
    e2 = (orot.matrix_to_axis_angle(orot.rotation_6d_to_matrix(d6_in)) - aa_out).abs().max().item()
    print("rotation oracle-vs-ref max abs", e1, e2)

    # ---- (ii-b) caller-side packing / long-form helpers ---------------------
    run_packing_goldens(ns)
    run_llm_parser_goldens(ns)
    run_gesture_type_goldens(ns)
    run_llm_retrieval_goldens(ns)
    if "--packing-only" in sys.argv:
        return

    # ---- (iii)-(v),(viii): model-level goldens --------------------------------------------
    with tempfile.TemporaryDirectory() as tmp:
        for tag, L, arch in (("L2_allenc", 2, "all_encoder"), ("L8_encdec", 8, "encoder_decoder")):
            cfg = synth.default_model_cfg(num_layers=L)
            vkw = dict(num_layers=4, ff_size=512) if arch == "encoder_decoder" else {}
            vae_cfgs = synth.synth_vae_cfgs(decoder_arch=arch, **vkw)
            model, full = build_reference_model(ns, cfg, vae_cfgs, seed=0, tmpdir=os.path.join(tmp, tag))
            if "--retrieval-only" not in sys.argv:
                run_model_goldens(ns, model, full, cfg, vae_cfgs, tag)
            if tag == "L2_allenc":
                run_retrieval_goldens(ns, model, tag)


def run_packing_goldens(ns):
    """tools/visualize.py:208-213, 266-291 and tools/longform_synthesis.py:262-265, 431-476, 714-741 are inline
    script code, not functions: the same call sequence is issued here on the REFERENCE's own rotation_conversions
    and torch ops, on seeded inputs; the body-part masks are built from the reference's joint table
    (mogen/datasets/utils/beatx_utils.py) exactly as beatx_dataset.py:82-109 builds them."""
    import importlib.util
    import torch.nn.functional as F
    from oracle import packing as opk
    spec = importlib.util.spec_from_file_location("beatx_utils", os.path.join(_ref_import.REF_ROOT, "mogen/datasets/utils/beatx_utils.py"))
    bu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bu)
    jl = bu.joints_list
    ori = jl["beat_smplx_joints"]
    masks = {}
    for part in ("upper", "lower", "hands", "face"):
        m = np.zeros(len(ori) * 3)
        for name in jl["beat_smplx_" + part]:
            m[ori[name][1] - ori[name][0]:ori[name][1]] = 1
        masks[part] = m
    om = opk.part_masks()
    for part in masks:
        assert np.array_equal(masks[part].astype(bool), om[part]), part
    rc = ns.rc
    g = np.random.Generator(np.random.PCG64(4242))
    u = lambda *sh: torch.from_numpy(g.uniform(-0.6, 0.6, size=sh).astype(np.float32))
    B, n = 1, 150
    up, lo, ha, fa = u(B, n, 39), u(B, n, 27), u(B, n, 90), u(B, n, 3)
    facial, trans = u(B, n, 100), u(B, n, 3)
    pred = torch.zeros(B, n, 165)
    pred[..., masks["upper"].astype(bool)] = up
    pred[..., masks["lower"].astype(bool)] = lo
    pred[..., masks["hands"].astype(bool)] = ha
    pred[..., masks["face"].astype(bool)] = fa

    def to30(m):
        bs, nn, dim = m.shape
        nj = dim // 3
        x = rc.matrix_to_rotation_6d(rc.axis_angle_to_matrix(m.reshape(bs, nn, nj, 3))).reshape(bs, nn, nj * 6)
        x = F.interpolate(x.permute(0, 2, 1), scale_factor=30 / 15, mode="linear").permute(0, 2, 1)
        return rc.matrix_to_axis_angle(rc.rotation_6d_to_matrix(x.reshape(bs, nn * 2, nj, 6))).reshape(bs, nn * 2, nj * 3)

    lin = lambda x: F.interpolate(x.permute(0, 2, 1), scale_factor=30 / 15, mode="linear").permute(0, 2, 1)
    poses30, facial30, trans30 = to30(pred), lin(facial), lin(trans)
    # long-form: two windows blended on 15 frames, then the final interpolation of the 285-frame motion
    ov = 15
    up2, lo2, ha2, fa2 = u(B, n, 39), u(B, n, 27), u(B, n, 90), u(B, n, 3)
    facial2, trans2 = u(B, n, 100), u(B, n, 3)
    pred2 = torch.zeros(B, n, 165)
    pred2[..., masks["upper"].astype(bool)] = up2
    pred2[..., masks["lower"].astype(bool)] = lo2
    pred2[..., masks["hands"].astype(bool)] = ha2
    pred2[..., masks["face"].astype(bool)] = fa2
    nj = 55
    keep_m, tail_m = pred[:, :-ov], pred[:, -ov:]
    m6 = rc.matrix_to_rotation_6d(rc.axis_angle_to_matrix(pred2.reshape(B, n, nj, 3))).reshape(B, n, nj * 6)
    t6 = rc.matrix_to_rotation_6d(rc.axis_angle_to_matrix(tail_m.reshape(B, ov, nj, 3))).reshape(B, ov, nj * 6)
    wn = torch.linspace(0, 1, ov).unsqueeze(0).unsqueeze(-1)
    wp = 1 - wn
    m6[:, :ov] = t6 * wp + m6[:, :ov] * wn
    f2, t2 = facial2.clone(), trans2.clone()
    f2[:, :ov] = facial[:, -ov:] * wp + f2[:, :ov] * wn
    t2[:, :ov] = trans[:, -ov:] * wp + t2[:, :ov] * wn
    m2 = rc.matrix_to_axis_angle(rc.rotation_6d_to_matrix(m6.reshape(B, n, nj, 6))).reshape(B, n, nj * 3)
    long_m = torch.cat([keep_m, m2], 1)
    long_f = torch.cat([facial[:, :-ov], f2], 1)
    long_t = torch.cat([trans[:, :-ov], t2], 1)
    assert long_m.shape[1] == 285
    long30 = to30(long_m)
    sample_len = 700
    starts = [0] + list(range(150 - 15, sample_len, 150 - 15))
    ends = [i + 150 for i in starts]
    np.savez(os.path.join(HERE, "packing.npz"), mask_upper=masks["upper"], mask_lower=masks["lower"], mask_hands=masks["hands"],
             mask_face=masks["face"], pred_motion=t2n(pred), poses30=t2n(poses30), facial30=t2n(facial30), trans30=t2n(trans30),
             long_motion=t2n(long_m), long_facial=t2n(long_f), long_trans=t2n(long_t), long30=t2n(long30),
             starts_700=np.array(starts), ends_700=np.array(ends))
    # oracle == reference
    e = lambda a, b: (a - b).abs().max().item()
    o_pred = opk.scatter_parts(up, lo, ha, fa)
    o_long = opk.blend_window(pred, facial, trans, pred2, facial2, trans2, ov)
    print("packing oracle-vs-ref max abs: scatter %g, to30 %g, lin %g, blend (m,f,t) %g %g %g, long30 %g; windows %s" % (
        e(o_pred, pred), e(opk.interp_motion(pred, 2), poses30), e(opk.interp_features(facial, 2), facial30),
        e(o_long[0], long_m), e(o_long[1], long_f), e(o_long[2], long_t), e(opk.interp_motion(o_long[0], 2), long30),
        opk.window_bounds(sample_len)[:2] == (starts, ends)))


LLM_OUTPUTS = [
    "[('hello', 'beat'), ('world', 'iconic')]",
    "[(\"over there\", \"deictic\"), (\"huge\", \"metaphoric\")]",
    "Here are the words:\n[('grow', 'metaphoric'), ('this', 'deictic')]\nExplanation: ('grow', 'metaphoric') because growth is abstract.",
    "1. \"spiral staircase\", iconic\n2. 'you', deictic\n3. so, beat",
    "[('up-and-down', 'iconic'), (\"don't\", 'metaphoric')]",
    "I could not find any gesture words.",
    "[('big', 'Metaphoric'), ('round', 'etaphoric'), ('tap', 'eat')]",
    "",
]


def run_llm_parser_goldens(ns):
    """rag/llm_retrieval.py:131-165 on answers in the formats the prompt asks for (and a few it does not)."""
    import json
    llm = importlib.import_module("mogen.models.transformers.rag.llm_retrieval")
    from oracle import retrieval as oret
    res = []
    for txt in LLM_OUTPUTS:
        ref = llm.parse_gesture_labels_from_llm_output(txt)
        assert oret.parse_gesture_labels_from_llm_output(txt) == ref, txt
        res.append(dict(llm_output=txt, labels=ref))
    with open(os.path.join(HERE, "llm_parser.json"), "w") as f:
        json.dump(res, f, indent=1)
    print("llm parser oracle == reference on %d answers" % len(res))


def f32_word_similarity(a, b):
    """Same stand-in, returned as numpy float32 like gensim's .similarity() does (the reference's scores then live
    in float32: NEP 50)."""
    return np.float32(synth.synth_word_similarity(a, b))


def run_gesture_type_goldens(ns):
    """rag/gesture_type_retrieval.py:8-176 on a synthetic labelled DB; the external word-embedding model behind
    get_word_similarity_score is replaced by synth.synth_word_similarity (python float) and by its float32 form."""
    import json
    from oracle import retrieval as oret
    gt = importlib.import_module("mogen.models.transformers.rag.gesture_type_retrieval")
    samples = synth.synth_retrieval_samples(200, seed=2025)
    db_text = {s["sample_name"]: (s["text_feature"], s["speaker_id"]) for s in samples}
    db_labels = {s["sample_name"]: [s["speaker_id"]] + s["gesture_labels"] for s in samples}   # raggesture.py:262
    assert oret.build_db_dicts(samples)["idx_2_gesture_labels"] == db_labels
    gold = {"queries": []}
    saved = gt.get_word_similarity_score
    try:
        for simname, sim in (("f64", synth.synth_word_similarity), ("f32", f32_word_similarity)):
            gt.get_word_similarity_score = sim
            for qseed in (11, 12, 13, 14, 15):
                q = synth.synth_query(qseed)
                labels = synth.synth_gesture_query(qseed, n_labels=2 + qseed % 2) if qseed != 14 else \
                    [dict(name="beat", word="and", start=0.2, end=0.5)]
                if qseed == 15:   # a type no DB entry carries (all scores 0 -> empty lists), an upper-case word
                    labels = [dict(name="emblem", word="peace", start=0.5, end=1.1),
                              dict(name="deictic", word="THAT ONE", start=2.0, end=3.4)]
                si, db_b, qb = gt.gesture_type_retrieval(text=None, gesture_labels=labels, speaker_id=q["speaker_id"],
                                                         db_idx_2_gesture_labels=db_labels,
                                                         encoded_text=q["text_features"], text_feat_cache=db_text)
                osi, odb, oqb = oret.gesture_type_retrieval(labels, q["speaker_id"], db_labels, q["text_features"], db_text, sim)
                assert osi == si and odb == db_b and oqb == qb, "oracle gesture_type retrieval != reference"
                gold["queries"].append(dict(seed=qseed, sim=simname, labels=labels,
                                            sample_indexes={str(k): v for k, v in si.items()},
                                            d_bounds={str(k): {n: list(b) for n, b in v.items()} for k, v in db_b.items()},
                                            query_bounds={str(k): list(v) for k, v in qb.items()}))
    finally:
        gt.get_word_similarity_score = saved
    with open(os.path.join(HERE, "gesture_type.json"), "w") as f:
        json.dump(gold, f, indent=1)
    print("gesture_type_retrieval oracle == reference on %d queries; result sizes:" % len(gold["queries"]),
          [[len(v) for v in q["sample_indexes"].values()] for q in gold["queries"]])


def run_llm_retrieval_goldens(ns):
    """rag/llm_retrieval.py:166-466 with get_llm_output replaced by the clip's canned answer (no network) and the
    word-similarity model by the deterministic stand-in; DB dicts built with the reference's own
    map_conns_to_prominence (raggesture.py:262-272)."""
    import json
    from oracle import retrieval as oret
    llm = importlib.import_module("mogen.models.transformers.rag.llm_retrieval")
    samples = synth.synth_retrieval_samples(200, seed=2025)
    db_text = {s["sample_name"]: (s["text_feature"], s["speaker_id"]) for s in samples}
    db_labels = {s["sample_name"]: [s["speaker_id"]] + s["gesture_labels"] for s in samples}
    db_gestprom = {s["sample_name"]: ns.rag_utils.map_conns_to_prominence([g["word"] for g in s["gesture_labels"]],
                                                                          s["prominence"]) for s in samples}
    mine = oret.build_db_dicts(samples)
    assert mine["idx_2_gestprom"] == db_gestprom
    assert sum(v is not None for d in db_gestprom.values() for v in d.values()) > 100
    gold = {"queries": []}
    saved = llm.get_word_similarity_score, llm.get_llm_output
    try:
        for simname, sim in (("f64", synth.synth_word_similarity), ("f32", f32_word_similarity)):
            llm.get_word_similarity_score = sim
            for qseed in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10):
                q = synth.synth_llm_query(qseed)
                if qseed == 7:    # the answer names a word the transcript does not contain, and a beat: nothing to retrieve
                    q["llm_output"] = "[(\"unicorn\", \"iconic\"), (\"%s\", \"beat\")]" % q["text_times"][0][1]
                if qseed == 8:    # no prominence known for the query words, speaker the DB does not have
                    q["prominence"] = [("zzz", 0.0, 0.1, 1.0)]
                    q["speaker_id"] = 99
                if qseed == 9:    # a three-word label that spans the clip's first words, quoted and upper-cased
                    ws = ["".join(c for c in t[1].lower() if c.isalnum() or c.isspace()) for t in q["text_times"][:3]]
                    q["llm_output"] = "1. \"%s\", deictic" % " ".join(ws).upper()
                if qseed == 10:   # the type names are matched case-sensitively (llm_retrieval.py:141): no label, empty result
                    q["llm_output"] = "(%s, Iconic)" % q["text_times"][1][1]
                answer = {q["text"]: q["llm_output"], "nothing here": "I cannot find any gesture words."}
                llm.get_llm_output = lambda t: answer[t]
                for text in ((q["text"], "   ") if qseed == 1 else (q["text"],)):
                    si, db_b, qb = llm.llm_retrieval(text=text, text_times=q["text_times"], speaker_id=q["speaker_id"],
                                                     prominence=q["prominence"], db_idx_2_gesture_labels=db_labels,
                                                     db_idx_2_prominence=db_gestprom, encoded_text=q["text_features"],
                                                     text_feat_cache=db_text)
                    osi, odb, oqb = oret.llm_retrieval(text, q["text_times"], q["speaker_id"], q["prominence"], db_labels,
                                                       mine["idx_2_gestprom"], q["text_features"], db_text, sim,
                                                       lambda t: answer[t])
                    assert osi == si and odb == db_b and oqb == qb, "oracle llm retrieval != reference"
                    gold["queries"].append(dict(seed=qseed, sim=simname, text=text, llm_output=q["llm_output"],
                                                text_times=[[list(t[0]), t[1]] for t in q["text_times"]],
                                                prominence=[list(p) for p in q["prominence"]], speaker_id=q["speaker_id"],
                                                sample_indexes={str(k): v for k, v in si.items()},
                                                d_bounds={str(k): {n: list(b) for n, b in v.items()} for k, v in db_b.items()},
                                                query_bounds={str(k): list(v) for k, v in qb.items()}))
    finally:
        llm.get_word_similarity_score, llm.get_llm_output = saved
    with open(os.path.join(HERE, "llm_retrieval.json"), "w") as f:
        json.dump(gold, f, indent=1)
    print("llm_retrieval oracle == reference on %d queries; result sizes:" % len(gold["queries"]),
          [[len(v) for v in q["sample_indexes"].values()] for q in gold["queries"]])


class _FakeDataset:
    """dataset[name] -> the per-sample dict RetrievalDatabase.forward reads (raggesture.py:558-570)."""

    def __init__(self, names):
        self.names = list(names)

    def __getitem__(self, key):
        name = self.names[0] if isinstance(key, int) else key
        import zlib
        d = synth.synth_batch(1, seed=zlib.crc32(name.encode()) & 0x7FFFFFFF)
        out = {k: d[k][0] for k in ("motion", "motion_upper", "motion_lower", "motion_face", "motion_hands", "facial",
                                    "trans", "contact", "motion_mask", "word", "audio")}
        out["speaker_id"] = d["speaker_ids"][0]
        out["sample_name"] = name
        return out


def run_retrieval_goldens(ns, model, tag):
    """Real reference: discourse_retrieval on a synthetic DB with forced ties, then the whole
    RetrievalDatabase.forward (selection, exemplar VAE encode with taped noise, placement)."""
    import json
    from oracle import retrieval as oret
    RD = ns.raggesture.RetrievalDatabase
    samples = synth.synth_retrieval_samples(200, seed=2025)
    names = [s["sample_name"] for s in samples]
    # DB dicts built with the REFERENCE's own map_conns_to_prominence (raggesture.py:255-276)
    db = dict(idx_2_text={}, idx_2_sense={}, idx_2_discbounds={}, idx_2_prominence={})
    for smp in samples:
        n, spk = smp["sample_name"], smp["speaker_id"]
        db["idx_2_text"][n] = (smp["text_feature"], spk)
        db["idx_2_sense"][n] = [spk] + [(d[1], d[0]) for d in smp["discourse"]]
        db["idx_2_discbounds"][n] = [(d[1], d[0], d[4], d[5], d[6], d[7]) for d in smp["discourse"]]
        db["idx_2_prominence"][n] = ns.rag_utils.map_conns_to_prominence([d[0] for d in smp["discourse"]], smp["prominence"])
    mine = oret.build_db_dicts(samples)
    assert mine["idx_2_prominence"] == db["idx_2_prominence"] and mine["idx_2_sense"] == db["idx_2_sense"]
    gold = {"queries": []}
    for qseed in (11, 12, 13, 14):
        q = synth.synth_query(qseed)
        si, db_b, qb = ns.discourse.discourse_retrieval(
            text=None, discourse=q["discourse"], prominence=q["prominence"], speaker_id=q["speaker_id"],
            db_idx_2_sense=db["idx_2_sense"], db_idx_2_discbounds=db["idx_2_discbounds"],
            db_idx_2_prominence=db["idx_2_prominence"], encoded_text=q["text_features"], text_feat_cache=db["idx_2_text"])
        osi, odb, oqb = oret.discourse_retrieval(q["discourse"], q["prominence"], q["speaker_id"], mine, q["text_features"])
        assert osi == si and odb == db_b and oqb == qb, "oracle retrieval != reference"
        gold["queries"].append(dict(seed=qseed, sample_indexes={str(k): v for k, v in si.items()},
                                    d_bounds={str(k): {n: list(b) for n, b in v.items()} for k, v in db_b.items()},
                                    query_bounds={str(k): list(v) for k, v in qb.items()}))
    print(tag, "discourse_retrieval oracle == reference on 4 queries; tie sizes:",
          [len(v) for v in gold["queries"][0]["sample_indexes"].values()])
    # ---- whole RetrievalDatabase.forward
    rdb = object.__new__(RD)
    torch.nn.Module.__init__(rdb)
    rdb.retrieval_method = {"discourse": ns.discourse.discourse_retrieval}
    rdb.idx_2_text, rdb.idx_2_sense = db["idx_2_text"], db["idx_2_sense"]
    rdb.idx_2_discbounds, rdb.idx_2_prominence = db["idx_2_discbounds"], db["idx_2_prominence"]
    rdb.train_indexes, rdb.train_dbounds, rdb.train_qbounds = {}, {}, {}
    rdb.test_indexes, rdb.test_dbounds, rdb.test_qbounds = {}, {}, {}
    rdb.num_retrieval, rdb.topk, rdb.max_seq_len, rdb.motion_framechunksize, rdb.motion_fps = 1, 2, 150, 15, 15
    rdb.latent_dim, rdb.text_latent_dim, rdb.dataset = 512, 768, _FakeDataset(names)
    rdb.eval()
    B = 2
    qs = [synth.synth_query(21), synth.synth_query(22)]
    data = synth.synth_batch(B, seed=99)
    own = ["query_clip_a", names[7]]  # second query pretends to be DB sample 7 (self-exclusion path)
    cond = dict(text=[None] * B, audio=[None] * B, text_enc=data["word"], text_features=[q["text_features"] for q in qs],
                audio_enc=data["audio"], discourse=[q["discourse"] for q in qs], prominence=[q["prominence"] for q in qs],
                speaker_ids=torch.tensor([[q["speaker_id"]] * 150 for q in qs]), gesture_labels=[[], []], text_times=[[], []])
    with torch.no_grad(), taped_noise(synth.NoiseTape(4242)):
        re = rdb(cond, [150] * B, "cpu", idx=own, retrieval_method="discourse", gesture_rep_encoder=model.model.gesture_rep_encoder)
    gold["forward"] = []
    lat = {}
    for b in range(B):
        ent = dict(own=own[b], retr_startends={str(k): list(v) for k, v in re["retr_startends"][b].items()},
                   query_startends={str(k): list(v) for k, v in re["query_startends"][b].items()},
                   names={str(k): v for k, v in re["raw_sample_names"][b].items()} if isinstance(re["raw_sample_names"][b], dict) else None)
        gold["forward"].append(ent)
        for k, v in re["retr_uncropped_latents"][b].items():
            lat["lat_%d_%d" % (b, k)] = t2n(v["retr_motion_latent"])
            lat["spk_%d_%d" % (b, k)] = t2n(v["retr_spkid"])
    gold["test_indexes"] = {n: {m: {str(k): v for k, v in d.items()} for m, d in e.items()} for n, e in rdb.test_indexes.items()}
    with open(os.path.join(HERE, "retrieval_%s.json" % tag), "w") as f:
        json.dump(gold, f, indent=1)
    np.savez(os.path.join(HERE, "retrieval_%s.npz" % tag), **lat)
    print(tag, "RetrievalDatabase.forward golden:", [(e["retr_startends"], e["query_startends"]) for e in gold["forward"]])


def run_model_goldens(ns, model, full, cfg, vae_cfgs, tag):
    from oracle import denoiser as od, diffusion as odf, vae as ovae, pipeline as opipe
    B = 2
    net = model.model
    res = {}
    # -- denoiser forward at 4 timesteps (CFG mix), real query masks and all-ones masks
    data = synth.synth_batch(B, seed=1234)
    g = np.random.Generator(np.random.PCG64(99))
    x = torch.from_numpy(g.standard_normal((B, 43, 512)).astype(np.float32))
    motion_mask = torch.ones(B, 43); motion_mask[:, [10, 21, 32]] = 0
    xf = od.encode_conditions(full, data["word"], data["audio"], data["speaker_ids"])
    qm_real = od.make_query_masks(motion_mask)
    qm_ones = {k: torch.ones_like(v) for k, v in qm_real.items()}
    with torch.no_grad():
        ref_xf = net.get_precompute_condition(text=data["word"], audio=data["audio"],
                                              speaker_ids=data["speaker_ids"], device="cpu", re_dict=1)["xf_out"]
        for k in xf:
            assert torch.allclose(xf[k], ref_xf[k], atol=1e-6), k
        for t in (999, 514, 99, 0):
            ts = torch.full((B,), t, dtype=torch.long)
            for mtag, qm in (("real", qm_real), ("ones", qm_ones)):
                ref = net(x, ts, motion_mask=motion_mask, xf_out=ref_xf, re_dict=1,
                          query_mask=copy.deepcopy(qm))
                mine = od.denoiser_forward(full, cfg, x, ts, motion_mask, xf, qm)
                keep = [r for r in range(43) if r not in (10, 20, 30)]
                err = (ref - mine).abs().max().item()
                errk = (ref[:, keep] - mine[:, keep]).abs().max().item()
                print(tag, "denoiser t=%d mask=%s oracle-vs-ref max abs %.3e (rows!=10,20,30: %.3e) |ref| %.3f"
                      % (t, mtag, err, errk, ref.abs().max().item()))
                res["den_%s_t%d" % (mtag, t)] = t2n(ref)
    np.savez(os.path.join(HERE, "denoiser_%s.npz" % tag), **res)

    # -- VAE encode (with explicit eps) / decode of a fixed latent
    sch = odf.SpacedSchedule()
    res = {}
    gre = net.gesture_rep_encoder
    data = synth.synth_batch(B, seed=1234)
    with torch.no_grad(), taped_noise(synth.NoiseTape(555)):
        lat_ref, mask_ref = gre.encode(data["motion_upper"], data["motion_lower"], data["motion_face"],
                                       data["motion_hands"], data["trans"].clone(), data["facial"],
                                       data["contact"], data["motion_mask"])
    tape = synth.NoiseTape(555)
    d2 = synth.synth_batch(B, seed=1234)
    lat_mine, mask_mine = ovae.gesture_encode(full, vae_cfgs, d2, [tape.draw((B * 10, 1, 512)) for _ in range(4)])
    print(tag, "vae encode oracle-vs-ref max abs %.3e |ref| %.3f" % ((lat_ref - lat_mine).abs().max().item(),
                                                                      lat_ref.abs().max().item()))
    assert torch.equal(mask_ref, mask_mine)
    res["enc_latent"] = t2n(lat_ref)
    zl = torch.from_numpy(g.standard_normal((B, 43, 512)).astype(np.float32))
    with torch.no_grad():
        dec_ref = gre.decode(zl)
    dec_mine = ovae.gesture_decode(full, vae_cfgs, zl)
    for nm, a, b in zip(("upper", "lower", "face", "hands", "transl", "exps", "contact"), dec_ref, dec_mine):
        print(tag, "vae decode %s oracle-vs-ref max abs %.3e |ref| %.3f" % (nm, (a - b).abs().max().item(), a.abs().max().item()))
        res["dec_" + nm] = t2n(a)
    np.savez(os.path.join(HERE, "vae_%s.npz" % tag), **res)

    # -- end to end: MotionDiffusion.forward(**data)
    e2e_B = 2 if cfg["num_layers"] <= 2 else 1
    gi = [0] * 25 + list(range(25))
    runs = [("base", dict(), False)]
    runs.append(("guided", dict(use_inversion=True, insertion_guidance=True, guidance_iters=gi, guidance_lr=0.1), True))
    if cfg["num_layers"] <= 2:
        runs.append(("invonly", dict(use_inversion=True), True))
        runs.append(("guidedprev", dict(use_inversion=True, insertion_guidance=True, guidance_iters=gi, guidance_lr=0.1,
                                        use_prev_latent=True), True))
        runs.append(("prevonly", dict(use_prev_latent=True), False))
    res = {}
    for rtag, ikw, need_re in runs:
        data = synth.synth_batch(e2e_B, seed=4321)
        re_dict = opipe.synthetic_re_dict(e2e_B, seed=77) if need_re else None
        prev = None
        if ikw.get("use_prev_latent"):
            prev = torch.from_numpy(np.random.Generator(np.random.PCG64(5)).standard_normal((e2e_B, 43, 512)).astype(np.float32))
        net.database = (lambda *a, **k: re_dict) if need_re else None
        rkw = dict(ikw)
        if prev is not None:
            rkw["prev_latent"] = prev.clone()
        with torch.no_grad(), taped_noise(synth.NoiseTape(2024)), contextlib.redirect_stdout(open(os.devnull, "w")):
            ref = model(**dict(data, retrieval_method="discourse", inference_kwargs=dict(rkw)))
        data2 = synth.synth_batch(e2e_B, seed=4321)
        okw = {k: v for k, v in ikw.items()}
        mine = opipe.motion_diffusion_forward(full, cfg, vae_cfgs, sch, data2, synth.NoiseTape(2024), re_dict=re_dict,
                                              prev_latent=prev.clone() if prev is not None else None, **okw)
        for k in ("prev_latentout", "pred_upper", "pred_lower", "pred_facepose", "pred_hands", "pred_transl", "pred_exps"):
            print(tag, rtag, k, "oracle-vs-ref max abs %.3e |ref| %.3f" % ((ref[k] - mine[k]).abs().max().item(),
                                                                            ref[k].abs().max().item()))
            res["%s_%s" % (rtag, k)] = t2n(ref[k])
    net.database = None
    np.savez(os.path.join(HERE, "e2e_%s.npz" % tag), **res)


if __name__ == "__main__":
    main()
