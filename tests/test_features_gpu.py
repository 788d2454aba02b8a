"""GPU: per-window conditioning feature extraction (tools/longform_synthesis.py:64-94) against the
Hugging Face implementations the reference calls (`transformers` is installed on the GPU box: BertModel /
Wav2Vec2Model at the bert-base-cased / wav2vec2-base-960h configurations, RANDOM-INITIALISED -- the released weights
cannot be fetched here -- evaluated in fp32 on the host as the oracle).
  BERT:      sum of the last four hidden states of the token sequence           [L, 768]
  wav2vec2:  last_hidden_state of a 10 s, 16 kHz window (processor normalisation)  [499, 768]
fp32 mode = bf16x3 GEMM operands (structural parity), bf16 = the production operands."""
import pytest
import torch

pytestmark = pytest.mark.gpu
transformers = pytest.importorskip("transformers")


def relerr(a, b):
    return ((a - b).norm() / b.norm()).item()


def test_bert_last_four_hidden_states(rg, parity):
    torch.manual_seed(0)
    cfg = transformers.BertConfig(vocab_size=28996)          # bert-base-cased: 12 layers, 768 wide, 12 heads, gelu, eps 1e-12
    model = transformers.BertModel(cfg, add_pooling_layer=False).eval()
    for L in (9, 47):
        ids = torch.randint(1000, 28996, (L,))
        ids[0], ids[-1] = 101, 102                            # [CLS] ... [SEP] as tokenizer.encode_plus produces
        with torch.no_grad():
            hs = model(input_ids=ids[None], output_hidden_states=True).hidden_states
        ref = torch.stack([hs[i] for i in (-4, -3, -2, -1)]).sum(0).squeeze(0)
        for precision, tol in (("fp32", 1e-4), ("bf16", 2e-2)):   # measured 7.7e-6 / 4.2e-3
            feats = rg.features.BertFeatures(model.state_dict(), device="cuda", precision=precision)
            got = feats(ids)
            torch.cuda.synchronize()
            st = feats.hidden_states(ids)
            assert len(st) == 13 and got.shape == (L, 768)
            e0 = relerr(st[0].cpu(), hs[0][0])
            e = relerr(got.cpu(), ref)
            parity.check("BERT-base L=%d %s: embeddings vs transformers" % (L, precision), e0, 1e-5)
            parity.check("BERT-base L=%d %s: sum of the last four layers vs transformers" % (L, precision), e, tol)


def test_wav2vec2_last_hidden_state(rg, parity):
    torch.manual_seed(1)
    cfg = transformers.Wav2Vec2Config()                       # wav2vec2-base: conv (512 x 7, group norm), 12 post-norm layers
    assert cfg.feat_extract_norm == "group" and not cfg.do_stable_layer_norm and cfg.num_conv_pos_embeddings == 128
    model = transformers.Wav2Vec2Model(cfg).eval()
    wave = (torch.randn(160000) * 0.05 + 0.01)
    xn = (wave - wave.mean()) / torch.sqrt(wave.var(unbiased=False) + 1e-7)     # Wav2Vec2FeatureExtractor(do_normalize=True)
    with torch.no_grad():
        ref = model(xn[None]).last_hidden_state[0]
        ref_conv = model.feature_extractor(xn[None])[0].T       # [499, 512] (Wav2Vec2Model returns them layer-normed)
    assert ref.shape == (499, 768)
    for precision, tol_conv, tol in (("fp32", 2e-4, 2e-3), ("bf16", 2e-2, 3e-2)):   # measured 1.4e-5, 3.8e-4 / 7.1e-3, 7.8e-3
        feats = rg.features.Wav2Vec2Features(model.state_dict(), device="cuda", precision=precision)
        conv = feats.conv_features(xn)
        got = feats(wave)
        torch.cuda.synchronize()
        assert conv.shape == (499, 512) and got.shape == (499, 768)
        ec, e = relerr(conv.cpu(), ref_conv), relerr(got.cpu(), ref)
        parity.check("wav2vec2-base %s: conv features vs transformers" % precision, ec, tol_conv)
        parity.check("wav2vec2-base %s: last hidden state vs transformers" % precision, e, tol)


def test_window_features_callback(rg):
    """longform_synthesis.py:320-343 as the long-form driver's `features` callback: audio slice of the window -> [1, 499, 768],
    merged transcript -> BERT sum of the last four layers (small random models: shapes and plumbing)."""
    torch.manual_seed(2)
    bert = transformers.BertModel(transformers.BertConfig(vocab_size=500, num_hidden_layers=4), add_pooling_layer=False).eval()
    w2v = transformers.Wav2Vec2Model(transformers.Wav2Vec2Config(num_hidden_layers=2)).eval()
    vocab = {}
    tok = lambda sentence: [101] + [vocab.setdefault(w, 110 + len(vocab)) for w in sentence.split()] + [102]
    wf = rg.features.WindowFeatures(rg.features.BertFeatures(bert.state_dict()), rg.features.Wav2Vec2Features(w2v.state_dict()), tok)
    raw = torch.randn(1, 16000 * 19) * 0.1                       # 19 s: the window [9, 19] s is whole, [18, 28] s is padded
    segs = [[[9.5, 9.9], "so"], [[9.5, 9.9], "me"], [[10.2, 10.8], "big"], [[11.0, 11.5], "house"]]
    f = wf.for_sample(raw)
    out = f(1, 9.0, 19.0, dict(text_segments=[segs]))
    assert out["raw_word"] == ["some big house"]                 # merge_disco_textsegs joins the two halves of "some"
    assert out["audio"].shape == (1, 499, 768) and out["text_features"][0].shape == (5, 768)
    ids = torch.tensor(tok("some big house"))
    with torch.no_grad():
        hs = bert(input_ids=ids[None], output_hidden_states=True).hidden_states
        wave = raw[0, 9 * 16000:19 * 16000]
        ref_a = w2v(((wave - wave.mean()) / torch.sqrt(wave.var(unbiased=False) + 1e-7))[None]).last_hidden_state
    assert relerr(out["text_features"][0].cpu(), torch.stack(hs[-4:]).sum(0)[0]) <= 2e-2
    assert relerr(out["audio"].cpu(), ref_a) <= 3e-2
    tail = f(2, 18.0, 28.0, dict(text_segments=[[]]))            # zero-padded audio tail, empty transcript -> [CLS] [SEP]
    assert tail["audio"].shape == (1, 499, 768) and tail["text_features"][0].shape == (2, 768) and tail["raw_word"] == [""]
