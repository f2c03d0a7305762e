"""Synthetic workloads for benchmarking without checkpoints or datasets (there is no network).

* ``humaneval_shaped_prompts``: token-id prompts with HumanEval-like lengths (SURVEY §8d config 3:
  clipped log-normal, median ~200, min 110, max 620), ids uniform below 151643.
* ``ScriptedAcceptance``: a logits hook that makes a random-init model behave, for the purposes of the
  Jacobi loop, like a Jacobi-Forcing-trained one: every absolute position has a target token; the
  prediction made at a position is the next target when the row's context is right, or — with
  probability ``robust`` — even when it is not (context robustness is what the training recipe
  instils, README.md:30-33); otherwise a hash-junk token.  Implemented as a few elementwise torch ops on
  the device plus one scatter into the logits, i.e. extra work inside the forward, never less.
"""
from __future__ import annotations

import math
from typing import List

import numpy as np
import torch

_M = (1 << 31) - 1


def humaneval_shaped_prompts(count: int, seed: int = 1234, vocab_hi: int = 151643) -> List[List[int]]:
    rng = np.random.default_rng(seed)
    lens = np.clip(np.exp(rng.normal(math.log(200.0), 0.45, size=count)), 110, 620).astype(np.int64)
    return [[int(t) for t in rng.integers(0, vocab_hi, size=int(L))] for L in lens]


def _hash(a: torch.Tensor, b: torch.Tensor, salt: int) -> torch.Tensor:
    x = (a * 1103515245 + b * 12345 + salt) & _M
    x = x ^ (x >> 15)
    x = (x * 48271) & _M
    x = x ^ (x >> 13)
    x = (x * 69621) & _M
    return x ^ (x >> 16)


class ScriptedAcceptance:
    def __init__(self, vocab: int, robust_pct: int = 75, seed: int = 7, boost: float = 1.0e4, vocab_hi: int = 151643):
        self.vocab, self.robust, self.seed, self.boost = vocab, int(robust_pct), int(seed), float(boost)
        self.vocab_hi = min(vocab_hi, vocab)

    def target(self, pos: torch.Tensor, prompt: torch.Tensor) -> torch.Tensor:
        return _hash(pos, prompt + self.seed, 0x9E37) % self.vocab_hi

    def __call__(self, logits: torch.Tensor, decoder, prefill=None) -> torch.Tensor:
        dev = logits.device
        if prefill is not None:
            p, plen = prefill
            n = logits.shape[0]                       # rows = positions plen-1 .. plen+n-2 of prompt ⧺ draft
            pos = torch.arange(plen - 1, plen - 1 + n, device=dev, dtype=torch.int64)
            pr = torch.full_like(pos, p)
            # the draft is random: only the last prompt position has a fully correct context
            ok = pos < plen
            rob = (_hash(pos, pr, 0x51ED) % 100) < self.robust
            nxt = torch.where(ok | rob, self.target(pos + 1, pr), _hash(pos, pr, 0x7777) % self.vocab_hi)
            logits[torch.arange(n, device=dev), nxt] = self.boost
            return logits
        b = decoder.batch
        R, T = b.Rtot, b.Tpad
        ids = b.input_ids[:R * T].view(R, T)
        pos = b.positions[:R * T].view(R, T).long()
        pr = b.row_prompt[:R].long().view(R, 1).expand(R, T)
        tgt = self.target(pos, pr)
        match = ids == tgt
        ok = torch.cumprod(match.to(torch.int32), dim=1).bool()          # committed prefix is always on-target
        rob = (_hash(ids, pos, 0x51ED) % 100) < self.robust
        nxt = torch.where(ok | rob, self.target(pos + 1, pr), _hash(ids, pos + pr, 0x7777) % self.vocab_hi)
        nxt = nxt.reshape(-1)
        vi = getattr(b, "valid_index", None)
        if vi is not None and logits.reshape(-1, logits.shape[-1]).shape[0] == vi.numel():
            nxt = nxt[vi.clamp(min=0).long()]               # logits of the compacted position list
        flat = logits.view(-1, logits.shape[-1])
        flat[torch.arange(flat.shape[0], device=dev), nxt] = self.boost
        return logits
