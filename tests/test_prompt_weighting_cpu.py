"""Prompt weighting (controlanimate_amd/prompt_weighting.py, the reference's Compel call at
modules/controlanimate_pipeline.py:133-135): the parser on the syntax the reference's configs use, and the embedding
arithmetic of the published algorithm on a stub tokenizer / encoder (Compel itself is not installable: parity unpinned)."""
import math
from types import SimpleNamespace

import pytest
import torch

from controlanimate_amd.prompt_weighting import Compel, parse_prompt


def test_parser_on_the_reference_config_syntax():
    # configs/prompts/SampleConfigLCM.yaml:16 (shortened)
    p = ("A woman with perfect++ face++ (female villain)+, (perfect face)++, (bad face)----, (worst quality)---- "
         "(plain bright golden room)+++ (muscle body)0.2, fully dressed, (gold pants)++ , high-quality, 8k")
    f = dict()
    frags = parse_prompt(p)
    for t, w in frags:
        f.setdefault(t, []).append(round(w, 4))
    assert frags[0] == ("A woman with", 1.0)
    assert f["perfect face"] == [1.21, 1.21]              # `perfect++ face++` merge; `(perfect face)++`
    assert f["female villain"] == [1.1]
    assert f["bad face"] == [round(0.9 ** 4, 4)] and f["plain bright golden room"] == [1.331]
    assert f["muscle body"] == [0.2] and f["gold pants"] == [1.21]
    assert frags[-1] == (", high-quality, 8k", 1.0)      # a hyphen inside a word is text
    # n_prompt of the same file: suffix followed by punctuation, a group holding commas
    n = parse_prompt("easynegative+, nudity, mask++, (text, font, logo)++, (nsfw,nude)+")
    assert n == [("easynegative", 1.1), (", nudity,", 1.0), ("mask", pytest.approx(1.21)), (",", 1.0),
                 ("text, font, logo", pytest.approx(1.21)), (",", 1.0), ("nsfw,nude", 1.1)]
    assert parse_prompt("(a (b)++ c)+") == [("a", 1.1), ("b", pytest.approx(1.1 * 1.21)), ("c", 1.1)]   # nesting multiplies
    assert parse_prompt(r"a \(literal\) b") == [("a (literal) b", 1.0)]
    assert parse_prompt("plain prompt, nothing weighted") == [("plain prompt, nothing weighted", 1.0)]
    with pytest.raises(NotImplementedError):
        parse_prompt('("a cat", "a dog").blend(0.5, 0.5)')


class Tok:
    """words -> ids (one id per word / punctuation mark), BOS 1, EOS = PAD 2."""
    model_max_length, bos_token_id, eos_token_id, pad_token_id = 12, 1, 2, 2

    def __init__(self):
        self.vocab = {}

    def __call__(self, text, **kw):
        import re
        ids = [self.vocab.setdefault(w, 10 + len(self.vocab)) for w in re.findall(r"\w+|[^\w\s]", text)]
        return SimpleNamespace(input_ids=[self.bos_token_id] + ids[: self.model_max_length - 2] + [self.eos_token_id])


def encoder(ids, attention_mask=None):
    """A causal toy encoder: position p sees the ids up to p (so removing a fragment changes everything after it);
    attention_mask [B, L] hides positions as KEYS, as transformers' CLIPTextModel does."""
    e = torch.sin(ids.float()[..., None] * torch.arange(1, 9).float() * 0.37)
    if attention_mask is None:
        return (torch.cumsum(e, 1) / torch.arange(1, ids.shape[1] + 1).float()[None, :, None],)
    m = attention_mask.float()[..., None]
    return (torch.cumsum(e * m, 1) / torch.cumsum(m, 1).clamp_min(1.0),)


def test_weighting_arithmetic():
    tok = Tok()
    c = Compel(tokenizer=tok, text_encoder=encoder)
    plain_ids, w = c.token_ids_and_weights([("a red cat", 1.0)])
    assert plain_ids.shape == (1, 12) and plain_ids[0, 0] == 1 and plain_ids[0, 4] == 2 and bool((plain_ids[0, 4:] == 2).all())
    z = encoder(plain_ids)[0]
    assert torch.equal(c("a red cat"), z)                                  # no weights: the plain encoding, bit for bit
    z0 = encoder(c.token_ids_and_weights([])[0])[0]
    up = c("a (red)++ cat")
    wt = torch.ones(1, 12, 1); wt[0, 2] = 1.21                             # BOS, a, red, cat, EOS, pad...
    assert torch.allclose(up, z0 + (z - z0) * wt, atol=1e-6)
    # w < 1, DownweightMode.REMOVE: blend with the prompt WITHOUT the fragment, tan((1 - w) pi / 2) : 1
    cr = Compel(tokenizer=tok, text_encoder=encoder, downweight_mode="remove")
    down = cr("a (red)0.5 cat")
    wt[0, 2] = 0.5
    base = z0 + (z - z0) * wt
    without = encoder(c.token_ids_and_weights([("a", 1.0), ("cat", 1.0)])[0])[0]
    t = math.tan(0.5 * math.pi / 2)
    assert torch.allclose(down, (base + t * without) / (1 + t), atol=1e-6)
    # w -> 0: the fragment is as good as removed
    assert torch.allclose(cr("a (red)0.0001 cat"), without, atol=1e-3)
    # w < 1, DownweightMode.MASK (Compel 2.0.2's default, and this class's): the SAME tokens with the fragment's positions
    # hidden in the attention mask, weighted per token like the base embedding
    assert c.downweight_mode == "mask"
    mask = torch.ones(1, 12, dtype=torch.long)
    mask[0, 2] = 0                                                         # "red"
    zm = encoder(plain_ids, attention_mask=mask)[0]
    masked = z0 + (zm - z0) * wt
    assert torch.allclose(c("a (red)0.5 cat"), (base + t * masked) / (1 + t), atol=1e-6)
    assert not torch.allclose(c("a (red)0.5 cat"), down, atol=1e-4)        # the two modes differ
    # a two-token fragment in the middle of the reference's sample prompt: positions 3..4 are hidden
    ranges = []
    c.token_ids_and_weights(parse_prompt("room (muscle body)0.2, fully dressed"), ranges)
    assert ranges[1] == (2, 4)
    # truncation to max_length - 2 tokens, weights cut with them
    ids, wts = c.token_ids_and_weights([("w1 w2 w3 w4 w5 w6", 1.0), ("w7 w8 w9 w10 w11 w12", 1.5)])
    assert ids.shape == (1, 12) and ids[0, -1] == 2 and wts[0].tolist() == [1.0] * 7 + [1.5] * 4 + [1.0]
    assert c(["a cat", "a (dog)+"]).shape == (2, 12, 8)
