"""Prompt weighting in the syntax the reference's configs are written in (`perfect++`, `(female villain)+`,
`(muscle body)0.2`, `(bad face)----`): what `compel_proc(self.prompt)` does at modules/controlanimate_pipeline.py:133-135.

Compel (`compel==2.0.2`, env.yml:115) is a third-party package that is not vendored in /root/reference and not installable
here, so this restates its published algorithm for the subset the reference uses; PARITY UNPINNED (no copy of the library
to compare with):

  syntax     `word+` / `word-`, `(phrase)+++`, `(phrase)1.3`, nesting multiplies; each `+` is x1.1, each `-` x0.9;
             `\\(` `\\)` escape literal parentheses.  `.and()`, `.blend()`, `.swap()` conjunctions are not implemented and
             raise (the reference's configs do not use them).
  tokens     every fragment is tokenised on its own, the ids are concatenated, cut to 75, wrapped in BOS / EOS and padded
             with the pad token; a token carries its fragment's weight, BOS / EOS / padding carry 1.
  weights    z = E(tokens), z0 = E(empty prompt):  weighted = z0 + (z - z0) * w_token   (so w = 1 everywhere is the plain
             encoding, bit for bit).
  w < 1      per down-weighted fragment a second embedding of the prompt "without" that fragment is blended in with weight
             tan((1 - w) * pi / 2) against 1 for the base embedding, normalised.  "Without" = Compel 2.0.2's default
             `DownweightMode.MASK` (round 3): the SAME token ids, with the fragment's token positions zeroed in the text
             encoder's attention mask (hidden as keys under the causal mask: ca_attention's key mask, ABI v7), weighted per
             token like the base embedding; `downweight_mode="remove"` re-tokenises the prompt without the fragment instead
             (`DownweightMode.REMOVE`).

Host-side string processing + a handful of text-encoder calls per window; the encoder itself is the HIP CLIPTextModel.
"""
from __future__ import annotations

import math
import re
from typing import Callable, List, Sequence, Tuple

import torch

Fragment = Tuple[str, float]

_CONJ = re.compile(r"\)\s*\.\s*(and|blend|swap)\s*\(")


def parse_prompt(text: str) -> List[Fragment]:
    """-> [(fragment text, weight)], adjacent fragments of equal weight merged, empty fragments dropped."""
    if _CONJ.search(text):
        raise NotImplementedError("prompt conjunctions (.and / .blend / .swap) are not implemented")
    pos = 0
    n = len(text)

    def suffix_weight() -> float:
        """A run of + / - or a number right after a `)` or a word: the multiplier it stands for."""
        nonlocal pos
        m = re.match(r"[+-]+(?![0-9.])|[+-]?(?:[0-9]+\.?[0-9]*|\.[0-9]+)", text[pos:])
        if not m:
            return 1.0
        s = m.group(0)
        pos += len(s)
        if s[-1] in "+-":
            return math.prod(1.1 if ch == "+" else 0.9 for ch in s)
        return float(s)

    def group(depth: int) -> List[Fragment]:
        nonlocal pos
        out: List[Fragment] = []
        buf = ""

        def flush():
            nonlocal buf
            if buf.strip():
                # a trailing run of + / - on the LAST word of the plain text weights that word only
                words = re.split(r"(\s+)", buf)
                plain = ""
                for w in words:
                    m = re.fullmatch(r"(.*?[^\s+\-,.;:!?])([+-]+)([,.;:!?]*)", w)
                    if m and not w.isspace():
                        if plain.strip():
                            out.append((plain, 1.0))
                        out.append((m.group(1), math.prod(1.1 if ch == "+" else 0.9 for ch in m.group(2))))
                        plain = m.group(3)  # punctuation after the suffix is plain text again
                    else:
                        plain += w
                if plain.strip():
                    out.append((plain, 1.0))
            buf = ""

        while pos < n:
            ch = text[pos]
            if ch == "\\" and pos + 1 < n and text[pos + 1] in "()":
                buf += text[pos + 1]
                pos += 2
            elif ch == "(":
                flush()
                pos += 1
                inner = group(depth + 1)
                w = suffix_weight()
                out.extend((t, fw * w) for t, fw in inner)
            elif ch == ")":
                if depth == 0:  # unbalanced: keep it as text, as a lenient parser would
                    buf += ch
                    pos += 1
                    continue
                flush()
                pos += 1
                return out
            else:
                buf += ch
                pos += 1
        flush()
        return out

    frags = group(0)
    merged: List[Fragment] = []
    for t, w in frags:
        t = " ".join(t.split())
        if not t:
            continue
        if merged and abs(merged[-1][1] - w) < 1e-9:
            merged[-1] = (merged[-1][0] + " " + t, w)
        else:
            merged.append((t, w))
    return merged


class Compel:
    """`Compel(tokenizer=..., text_encoder=...)(prompt) -> [1, 77, dim]` (the reference's call, :133-135)."""

    def __init__(self, tokenizer, text_encoder: Callable, truncate_long_prompts: bool = True, device=None, downweight_mode: str = "mask"):
        self.tokenizer, self.text_encoder, self.device = tokenizer, text_encoder, device
        if downweight_mode not in ("mask", "remove"):
            raise ValueError("downweight_mode must be 'mask' (Compel's default) or 'remove'")
        self.downweight_mode = downweight_mode
        if not truncate_long_prompts:
            raise NotImplementedError("only truncate_long_prompts=True (the reference's default) is implemented")
        self.max_length = int(getattr(tokenizer, "model_max_length", 77))

    # -- tokens ---------------------------------------------------------------------------------------------------
    def _fragment_ids(self, text: str) -> List[int]:
        ids = self.tokenizer(text, truncation=True, max_length=self.max_length, padding="do_not_pad").input_ids
        ids = list(ids[0]) if ids and isinstance(ids[0], (list, tuple)) else list(ids)
        return ids[1:-1]  # without BOS / EOS

    def token_ids_and_weights(self, fragments: Sequence[Fragment], ranges: list = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """ranges (optional list): receives per fragment the [start, end) token positions it occupies in the 77-token
        sequence (after the cut to 75 content tokens; BOS is position 0)."""
        tok = self.tokenizer
        ids: List[int] = []
        wts: List[float] = []
        room = self.max_length - 2
        for text, w in fragments:
            f = self._fragment_ids(text)
            if ranges is not None:
                ranges.append((1 + min(len(ids), room), 1 + min(len(ids) + len(f), room)))
            ids += f
            wts += [w] * len(f)
        ids, wts = ids[:room], wts[:room]
        pad = tok.pad_token_id if getattr(tok, "pad_token_id", None) is not None else tok.eos_token_id
        npad = room - len(ids)
        ids = [tok.bos_token_id] + ids + [tok.eos_token_id] + [pad] * npad
        wts = [1.0] + wts + [1.0] + [1.0] * npad
        return torch.tensor([ids], dtype=torch.long), torch.tensor([wts], dtype=torch.float32)

    # -- embeddings -----------------------------------------------------------------------------------------------
    def _encode(self, ids: torch.Tensor, mask: torch.Tensor = None) -> torch.Tensor:
        if self.device is not None:
            ids = ids.to(self.device)
            mask = None if mask is None else mask.to(self.device)
        if mask is None:
            return self.text_encoder(ids)[0].float()
        return self.text_encoder(ids, attention_mask=mask)[0].float()

    def _weighted(self, fragments: Sequence[Fragment], hide: Tuple[int, int] = None) -> torch.Tensor:
        """z0 + (z - z0) * w per token.  hide = (start, end): those token positions are invisible as attention keys."""
        ids, w = self.token_ids_and_weights(fragments)
        mask = None
        if hide is not None and hide[1] > hide[0]:
            mask = torch.ones_like(ids)
            mask[0, hide[0]:hide[1]] = 0
        z = self._encode(ids, mask)
        if bool((w == 1.0).all()):
            return z
        z0 = self._encode(self.token_ids_and_weights([])[0])
        return z0 + (z - z0) * w.to(z.device)[..., None]

    def __call__(self, text) -> torch.Tensor:
        if isinstance(text, (list, tuple)):
            return torch.cat([self(t) for t in text])
        fragments = parse_prompt(text)
        embeddings = [self._weighted(fragments)]
        lerp = [1.0]
        ranges: list = []
        self.token_ids_and_weights(fragments, ranges)
        for i, (_, w) in enumerate(fragments):
            if w < 1.0:
                if self.downweight_mode == "mask":  # Compel 2.0.2's default: same tokens, the fragment hidden in the attention
                    embeddings.append(self._weighted(fragments, hide=ranges[i]))
                else:
                    embeddings.append(self._weighted(list(fragments[:i]) + list(fragments[i + 1:])))
                lerp.append(math.tan((1.0 - max(1e-5, w)) * math.pi / 2))
        if len(embeddings) == 1:
            return embeddings[0]
        tot = sum(lerp)
        return sum(e * (l / tot) for e, l in zip(embeddings, lerp))
