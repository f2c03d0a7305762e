"""Scripted deterministic "language model" used by the oracle, the golden-vector
generator and the parity tests.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the product
package ``jacobiforcing_amd``; only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may use it.

The model is a pure function of the *whole causal context* of a row, so a wrong
KV commit / wrong candidate row / wrong trim shows up as diverging integers:

* every absolute position ``p >= prompt_len`` has a "true" token ``target(p)``;
  greedy autoregressive decoding from the prompt yields exactly
  ``target(prompt_len), target(prompt_len+1), ...`` (the reference's own greedy
  criterion: Jacobi output == AR output, inference_engine/tests/
  test_jacobi_decoding_greedy.py:180-206);
* the prediction made at position ``i`` (for token ``i+1``) is ``target(i+1)`` when
  every generated token at positions ``prompt_len..i`` equals its target, or — with
  probability ``robust_pct`` %, decided by a hash of the rolling context hash —
  anyway ("context-robust", the property Jacobi-Forcing training instils);
  otherwise it is a junk token derived from the rolling context hash.

Everything is integer arithmetic on Python ints / numpy uint64, no torch.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

_M32 = 0xFFFFFFFF


def mix32(a: int, b: int) -> int:
    """Small avalanche hash of two ints -> uint32 (deterministic everywhere)."""
    x = ((a & _M32) * 0x9E3779B1 + (b & _M32) * 0x85EBCA77 + 0x27D4EB2F) & _M32
    x ^= x >> 15
    x = (x * 0x2C1B3C6D) & _M32
    x ^= x >> 12
    x = (x * 0x297A2D39) & _M32
    x ^= x >> 15
    return x


class ScriptedModel:
    """Deterministic causal predictor over a vocabulary of ``vocab`` ids.

    Parameters
    ----------
    vocab:       vocabulary size V (token ids 0..V-1)
    seed:        selects the target sequence
    robust_pct:  0..100, chance a prediction is right despite a wrong context
    prompt_len:  positions < prompt_len are prompt (never "wrong")
    eos_id / eos_pos: when given, ``target(eos_pos) == eos_id`` (absolute position)
    reserved:    ids never produced by target()/junk (e.g. pad id, eos id)
    period:      when > 0 the target sequence repeats with this period (forces
                 n-gram-pool hits for rejection recycling)
    peak:        logit planted at the greedy id by logits_rows() (noise is in (-1, 1)); 8.0 makes the
                 softmax nearly one-hot, smaller values spread the mass (non-greedy verify cases)
    """

    def __init__(self, vocab: int, seed: int, robust_pct: int, prompt_len: int,
                 eos_id: Optional[int] = None, eos_pos: Optional[int] = None,
                 reserved: Sequence[int] = (), period: int = 0, peak: float = 8.0):
        self.vocab = int(vocab)
        self.seed = int(seed)
        self.robust_pct = int(robust_pct)
        self.prompt_len = int(prompt_len)
        self.eos_id = eos_id
        self.eos_pos = eos_pos
        self.reserved = set(int(r) for r in reserved)
        if eos_id is not None:
            self.reserved.add(int(eos_id))
        self.period = int(period)
        self.peak = float(peak)
        self._free = [t for t in range(self.vocab) if t not in self.reserved]
        assert len(self._free) >= 2

    # -- token rules -------------------------------------------------------
    def _pick(self, h: int) -> int:
        return self._free[h % len(self._free)]

    def target(self, pos: int) -> int:
        if self.eos_pos is not None and pos == self.eos_pos:
            return int(self.eos_id)
        p = pos % self.period if self.period > 0 else pos
        return self._pick(mix32(self.seed, p))

    def prompt(self) -> List[int]:
        return [self._pick(mix32(self.seed ^ 0x5151, p)) for p in range(self.prompt_len)]

    # -- causal prediction ---------------------------------------------------
    def ctx_init(self, committed: Sequence[int]):
        """Fold a committed token list (prompt + accepted tokens, positions
        0..len-1) into (rolling hash, all-correct flag)."""
        h, ok = self.seed & _M32, True
        for pos, tok in enumerate(committed):
            h, ok = self.ctx_step(h, ok, pos, tok)
        return h, ok

    def ctx_step(self, h: int, ok: bool, pos: int, tok: int):
        h = mix32(h, (tok << 1) ^ pos)
        if pos >= self.prompt_len and tok != self.target(pos):
            ok = False
        return h, ok

    def predict(self, h: int, ok: bool, pos: int) -> int:
        """Greedy next-token prediction made AT position ``pos`` (for pos+1)
        given the folded context up to and including pos."""
        if ok or (mix32(h, pos ^ 0xABCD) % 100) < self.robust_pct:
            return self.target(pos + 1)
        return self._pick(mix32(h, 0x7777 + pos))

    def greedy_rows(self, committed: Sequence[int], rows: Sequence[Sequence[int]]) -> List[List[int]]:
        """Predictions for every position of every row; all rows continue the
        same committed prefix (the reference forwards candidate rows over one
        shared prefix KV, MB:421-436)."""
        h0, ok0 = self.ctx_init(committed)
        base = len(committed)
        out = []
        for row in rows:
            h, ok = h0, ok0
            g = []
            for t, tok in enumerate(row):
                h, ok = self.ctx_step(h, ok, base + t, int(tok))
                g.append(self.predict(h, ok, base + t))
            out.append(g)
        return out

    def logits_rows(self, committed: Sequence[int], rows: Sequence[Sequence[int]],
                    dtype=np.float32) -> np.ndarray:
        """Dense logits [B, T, V]: hash noise in (-1, 1) plus ``peak`` (8.0) at the greedy id."""
        g = self.greedy_rows(committed, rows)
        B, T = len(rows), (len(rows[0]) if rows else 0)
        lg = np.empty((B, T, self.vocab), dtype=np.float32)
        base = len(committed)
        v = np.arange(self.vocab, dtype=np.uint64)
        for b in range(B):
            for t in range(T):
                s = np.uint64(mix32(self.seed + 17 * b, base + t))
                x = (v * np.uint64(2654435761) + s * np.uint64(40503)) & np.uint64(0xFFFF)
                lg[b, t] = x.astype(np.float32) / np.float32(32768.0) - np.float32(1.0)
                lg[b, t, g[b][t]] = np.float32(self.peak)
        return lg.astype(dtype)

    def ar_continuation(self, start: int, count: int) -> List[int]:
        return [self.target(p) for p in range(start, start + count)]

    def describe(self) -> dict:
        d = dict(vocab=self.vocab, seed=self.seed, robust_pct=self.robust_pct,
                 prompt_len=self.prompt_len, eos_id=self.eos_id, eos_pos=self.eos_pos,
                 reserved=sorted(self.reserved - ({self.eos_id} if self.eos_id is not None else set())),
                 period=self.period)
        if self.peak != 8.0:          # recorded only when it differs, so older fixtures keep their descriptors
            d["peak"] = self.peak
        return d

    @classmethod
    def from_dict(cls, d: dict) -> "ScriptedModel":
        return cls(**d)
