"""On-policy rollout records: the API, record format and semantics of the reference's
``JacobiDecoderNonGreedyOnPolicy`` (inference_engine/engine/jacobi_decoding_nongreedy_on_policy.py = "JDO").

OUTPUT FORMAT (JDO:7-28): ``generate_rollout_records_batch(seqs)`` returns one dict per sequence keyed by block index;
each block record holds ``diffusion_itr_id`` ("itr_k"), ``data_id``, ``prompt_ids`` (prefix before the block, left padding
trimmed), ``answer_trajectory_ids`` (block-local trajectory: one fixed-length vector per forward = accepted prefix + current
drafted suffix), ``teacher_output_ids`` (final prompt + completion, stop-truncated, filled in for ALL blocks at the end) and
the cumulative ``tokens_per_iter / tokens_per_forward / num_iters / num_forwards``.

Per forward the verify (accept iff u < p, bonus != proposed on the first rejection, stop-token set; JDO:270-327) and the
fresh multinomial sample of every not yet accepted position (JDO:465-477) run on the GPU: ``jf_rs_probs`` reads the logits
once, ``jf_rs_onpolicy_step`` does the sequential part and the per-row inverse-CDF draws; one read-back per forward instead
of two ``.item()`` syncs per position.  Randomness comes from pre-drawn streams consumed in the reference's order
(``torch.rand`` JDO:306, ``torch.multinomial`` JDO:154, ``random.choice`` JDO:266); ``set_streams`` injects them.
"""
from __future__ import annotations

import os
import random
from typing import Dict, List, Optional, Sequence as PySeq, Tuple, Union

import numpy as np
import torch
from torch import Tensor

from .. import ops
from .block_manager import BlockManager
from .jacobi_decoding import LogitsForwardFn, LogitsForwardFnBatch
from .sequence import Sequence

_STREAM_LEN = 1 << 16


_DATA_ID_ATTRS = ("data_id", "request_id", "req_id", "uid", "id")        # the request attributes JDO:69-74 looks at, in order


def _infer_data_id(seq, fallback: str) -> str:
    """First request attribute of _DATA_ID_ATTRS that is set, as a string; else ``fallback``."""
    found = next((getattr(seq, a) for a in _DATA_ID_ATTRS if getattr(seq, a, None) is not None), None)
    return fallback if found is None else str(found)


def _trim_left_padding(ids: List[int], pad_token_id: Optional[int]) -> List[int]:
    """Drop the run of pad tokens a left-padded prompt starts with (JDO:77-87); an all-pad list becomes empty."""
    ids = list(ids)
    if pad_token_id is None:
        return ids
    keep = next((k for k, tok in enumerate(ids) if int(tok) != int(pad_token_id)), len(ids))
    return ids[keep:]


def _truncate_after_stop(ids: List[int], start_idx: int, stop_ids: PySeq[int]) -> List[int]:
    """Cut after the first stop token at or behind ``start_idx`` (the prompt itself may contain stop ids, JDO:170-182)."""
    ids, stop = list(ids), frozenset(int(x) for x in stop_ids)
    first = max(0, int(start_idx))
    hit = next((k for k in range(first, len(ids)) if int(ids[k]) in stop), None)
    return ids if hit is None else ids[:hit + 1]


def _require(value, what: str):
    """The constructor's contract (JDO:205-221): ids and sizes come from the model config, never from defaults."""
    if value is None:
        raise ValueError(f"{what} must be provided from model config. Do not use hard-coded values.")
    return value


class _BlockState:
    """The fixed-length vector of one block while it is being decoded: accepted prefix ⧺ current guesses ⧺ PAD beyond the
    token budget, and the trajectory of its snapshots (one per forward, plus the initial guess)."""

    def __init__(self, full_len: int, gen_len: int, pad: int, stop_ids):
        self.full_len, self.gen_len, self.pad, self.stop = full_len, gen_len, int(pad), set(int(x) for x in stop_ids)
        self.tokens: List[int] = []
        self.accepted = 0
        self.stopped = False
        self.trajectory: List[List[int]] = []

    def start(self, init: List[int]) -> None:
        self.tokens = list(init) + [self.pad] * (self.full_len - self.gen_len)
        self.snapshot()

    def open(self) -> bool:
        return self.accepted < self.gen_len and not self.stopped

    def pending(self) -> List[int]:
        return [int(t) for t in self.tokens[self.accepted:self.gen_len]]

    def accept(self, committed: List[int]) -> int:
        """Write the committed tokens over the guesses; returns the first position they went to (JDO:433-436)."""
        first = self.accepted
        self.accepted = min(self.gen_len, first + len(committed))
        self.tokens[first:self.accepted] = committed[:self.accepted - first]
        return first

    def close_at_stop(self, first_new: int) -> None:
        """A stop token was committed: PAD everything after it (JDO:452-463)."""
        self.stopped = True
        pos = next((j for j in range(first_new, self.accepted) if int(self.tokens[j]) in self.stop), None)
        if pos is not None:
            self.tokens[pos + 1:] = [self.pad] * (self.full_len - pos - 1)
            self.accepted = min(self.accepted, pos + 1)

    def refill(self, samples: List[int]) -> None:
        need = self.gen_len - self.accepted
        self.tokens[self.accepted:self.gen_len] = [int(t) for t in samples[:need]]

    def snapshot(self) -> None:
        self.tokens[self.gen_len:] = [self.pad] * (self.full_len - self.gen_len)
        self.trajectory.append(list(self.tokens))


class JacobiDecoderNonGreedyOnPolicy:
    def __init__(self, block_manager: BlockManager, forward_step: Optional[LogitsForwardFn] = None,
                 forward_step_batch: Optional[LogitsForwardFnBatch] = None,
                 eos_token_id: Optional[Union[int, List[int], Tuple[int, ...], set]] = None,
                 pad_token_id: Optional[int] = None, vocab_size: Optional[int] = None,
                 device: Optional[torch.device] = None) -> None:
        if forward_step is None and forward_step_batch is None:
            raise ValueError("Provide at least one of forward_step or forward_step_batch.")        # JDO:196-197
        self.block_manager = block_manager
        self.forward_step = forward_step
        self.forward_step_batch = forward_step_batch
        eos = _require(eos_token_id, "eos_token_id")                                              # JDO:205-210
        many = isinstance(eos, (list, tuple, set))
        self.stop_token_ids = tuple(int(t) for t in (eos if many else (eos,)))
        self.pad_token_id = int(_require(pad_token_id, "pad_token_id"))
        self.vocab_size = int(_require(vocab_size, "vocab_size"))
        self.debug = os.environ.get("JACOBI_DEBUG", "0") == "1"
        if device is None:
            device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.device = torch.device(device)
        self._rnd = random
        self._unis = self._multi = None
        self._cur = [0, 0]               # uniforms, multinomial draws consumed
        self._stepper: Optional[ops.OnPolicyStepper] = None

    # ------------------------------------------------------------------------------- randomness
    def set_streams(self, rnd, uniforms, multinomial) -> None:
        """``rnd``: object with ``choice(seq)`` / ``randrange(n)`` (draft initialisation); ``uniforms``: floats in [0,1)
        for the accept tests; ``multinomial``: floats in [0,1) turned into samples by inverse CDF (bonus + re-draft)."""
        self._rnd = rnd
        self._unis = np.asarray(uniforms, dtype=np.float32).copy()
        self._multi = np.asarray(multinomial, dtype=np.float32).copy()
        self._cur = [0, 0]
        self._stepper = None

    def _ensure(self, L: int) -> ops.OnPolicyStepper:
        if self._unis is None:
            self._unis = torch.rand(_STREAM_LEN).numpy()
            self._multi = torch.rand(_STREAM_LEN).numpy()
        st = self._stepper
        if st is None or st.max_L < L:
            st = ops.OnPolicyStepper(max(L, 64 if st is None else st.max_L), self.device, self._unis, self._multi,
                                     self.stop_token_ids)
            self._stepper = st
        return st

    # ------------------------------------------------------------------------------- config helpers
    def _get_sampling_cfg(self, seq: Sequence) -> Tuple[int, int, int]:
        """(block_len, max_blocks, remaining budget); ``jacobi_max_iterations`` is the max number of BLOCKS (JDO:228-244)."""
        sp = getattr(seq, "sampling_params", None)
        g = lambda name, default: getattr(sp, name, default) if sp is not None else default
        remaining = max(0, int(g("max_tokens", 2048)) - int(getattr(seq, "num_completion_tokens", 0)))
        return int(g("jacobi_block_len", 64)), int(g("jacobi_max_iterations", 128)), remaining

    @torch.inference_mode()
    def _forward_single(self, seq: Sequence, draft: Tensor) -> Tensor:
        if self.forward_step is not None:
            return self.forward_step(seq, draft)
        return self.forward_step_batch([seq], draft)

    def _init_block_draft_from_prompt(self, prompt_ids: List[int], block_len: int) -> List[int]:
        """JDO:254-266: sample from the prompt's tokens with replacement, pads excluded."""
        if block_len <= 0:
            return []
        pad = self.pad_token_id
        choices = [int(t) for t in prompt_ids if int(t) != pad]
        if not choices:
            return [self._rnd.randrange(self.vocab_size) for _ in range(block_len)]
        return [self._rnd.choice(choices) for _ in range(block_len)]

    # ------------------------------------------------------------------------------- one block (JDO:331-488)
    @torch.inference_mode()
    def _run_one_block(self, seq: Sequence, block_len: int, token_budget_remaining: int, completion_start_len: int,
                       profiler=None) -> Tuple[List[List[int]], int, int, bool]:
        """Returns (trajectory, tokens appended, forwards used, stopped)."""
        if int(block_len) <= 0 or token_budget_remaining <= 0:
            return [], 0, 0, True
        blk = _BlockState(int(block_len), min(int(block_len), int(token_budget_remaining)), self.pad_token_id,
                          self.stop_token_ids)
        blk.start(self._init_block_draft_from_prompt(list(seq.token_ids), blk.gen_len))
        sp = getattr(seq, "sampling_params", None)
        temperature = float(getattr(sp, "temperature", 1.0)) if sp is not None else 1.0
        top_k, top_p = ops.active_filters(sp, self.vocab_size)         # planted on the request object (JDO:132-133 reads them with getattr)
        st = self._ensure(blk.full_len + 1)
        bm = self.block_manager
        tick = (lambda name, on: (profiler.start(name) if on else profiler.stop(name))) if profiler else (lambda name, on: None)
        fwd_used = appended_total = 0
        while blk.open():
            if not seq.token_ids:
                seq.token_ids = [self.pad_token_id]
            proposed = blk.pending()
            remaining = len(proposed)
            draft = torch.tensor([[int(seq.token_ids[-1])] + proposed], dtype=torch.long, device=self.device)
            seq.draft_tokens = draft[0].tolist()
            tick("jacobi.forward", True)
            logits = self._forward_single(seq, draft)                               # [1, remaining, V]
            fwd_used += 1
            tick("jacobi.forward", False)
            if logits.ndim != 3 or int(logits.size(0)) != 1 or int(logits.size(1)) != remaining:
                raise ValueError(f"forward must return logits [1, {remaining}, vocab], got {tuple(logits.shape)}")
            V = int(logits.size(-1))
            bad = next((x for x in proposed if not 0 <= x < V), None)                # JDO:298-301
            if bad is not None:
                raise ValueError(f"Token index {bad} out of bounds for vocab size {V}. "
                                 "This may indicate a mismatch between model vocab and tokenizer vocab.")
            tick("jacobi.verify", True)
            row, committed, redraft = st.step(draft[0, 1:], logits[0], temperature, self._cur, top_k, top_p)
            self._cur[0] += row["n_uniforms"]
            self._cur[1] += row["n_bonus_draws"] + row["n_redraft"]
            stop_hit = bool(row["stop_hit"])
            tick("jacobi.verify", False)
            if not committed:
                committed, stop_hit = [proposed[0]], proposed[0] in blk.stop
            tick("jacobi.commit", True)
            seq.extend_tokens(committed)                                            # JDO:412-416
            if bm is not None:
                bm.may_append_batch(seq, len(committed))
                if remaining > len(committed):
                    bm.trim_kv_only_fast(seq, remaining - len(committed))           # JDO:420-426
            tick("jacobi.commit", False)
            appended_total += len(committed)
            seq.clear_draft()
            if len(seq) != seq.num_cached_tokens:
                raise RuntimeError(f"Invariant violated: len(token_ids)={len(seq)} != num_cached_tokens={seq.num_cached_tokens}")
            first_new = blk.accept(committed)
            if stop_hit:                                                             # JDO:439-463
                kept = _truncate_after_stop(list(seq.token_ids), completion_start_len, self.stop_token_ids)
                dropped = len(seq.token_ids) - len(kept)
                if dropped:
                    seq.token_ids = kept
                    if bm is not None:
                        bm.trim_kv_only_fast(seq, dropped)
                blk.close_at_stop(first_new)
            elif blk.accepted < blk.gen_len:                                         # JDO:465-477 (samples drawn on the GPU)
                blk.refill(redraft[len(committed):])
            blk.snapshot()
        return blk.trajectory, appended_total, fwd_used, blk.stopped

    # ------------------------------------------------------------------------------- records (JDO:494-614)
    @torch.inference_mode()
    def generate_rollout_records_batch(self, seqs: List[Sequence], n_token_seq_len: Optional[int] = None,
                                       return_metrics: bool = False) -> object:
        if not seqs:
            return ([], []) if return_metrics else []
        B = len(seqs)
        completion_start_lens = [len(list(s.token_ids)) for s in seqs]
        cfgs = [self._get_sampling_cfg(s) for s in seqs]
        block_lens = [int(n_token_seq_len) if n_token_seq_len is not None else int(c[0]) for c in cfgs]
        max_blocks = [int(c[1]) for c in cfgs]
        budgets = [int(c[2]) for c in cfgs]
        stopped = [False] * B
        num_blocks_done, num_forwards, total_generated = [0] * B, [0] * B, [0] * B
        data_ids = [_infer_data_id(seqs[i], fallback=f"data_{i}") for i in range(B)]
        per_seq_out: List[Dict[int, Dict[str, object]]] = [dict() for _ in range(B)]
        while True:
            active = [i for i in range(B) if not stopped[i] and num_blocks_done[i] < max_blocks[i] and budgets[i] > 0]
            if not active:
                break
            for i in active:
                seq, k = seqs[i], int(num_blocks_done[i])
                if block_lens[i] <= 0:
                    stopped[i] = True
                    continue
                prompt_ids_trim = _trim_left_padding(list(seq.token_ids), self.pad_token_id)
                traj, appended_now, fwd_used, stop_hit = self._run_one_block(seq, block_lens[i], budgets[i],
                                                                             completion_start_lens[i])
                num_blocks_done[i] += 1
                num_forwards[i] += int(fwd_used)
                total_generated[i] += int(appended_now)
                budgets[i] = max(0, budgets[i] - int(appended_now))
                stopped[i] = bool(stop_hit)
                after = _truncate_after_stop(list(seq.token_ids), completion_start_lens[i], self.stop_token_ids)
                tok = float(total_generated[i])
                per_seq_out[i][k] = {
                    "diffusion_itr_id": f"itr_{k}", "data_id": str(data_ids[i]), "prompt_ids": prompt_ids_trim,
                    "answer_trajectory_ids": traj, "teacher_output_ids": _trim_left_padding(after, self.pad_token_id),
                    "tokens_per_iter": tok / float(max(1, num_blocks_done[i])),
                    "tokens_per_forward": tok / float(max(1, num_forwards[i])),
                    "num_iters": int(num_blocks_done[i]), "num_forwards": int(num_forwards[i]),
                }
        final_teacher_by_id: Dict[str, List[int]] = {}
        for i in range(B):
            full = _truncate_after_stop(list(seqs[i].token_ids), completion_start_lens[i], self.stop_token_ids)
            final_teacher_by_id[str(data_ids[i])] = _trim_left_padding(full, self.pad_token_id)
        for i in range(B):
            for k in list(per_seq_out[i].keys()):
                per_seq_out[i][k]["teacher_output_ids"] = final_teacher_by_id.get(str(data_ids[i]), [])
        if not return_metrics:
            return per_seq_out
        metrics = []
        for i in range(B):
            it, fw, tok = float(max(1, num_blocks_done[i])), float(max(1, num_forwards[i])), float(total_generated[i])
            metrics.append({"total_tokens": tok, "num_iters": float(num_blocks_done[i]),
                            "num_forwards": float(num_forwards[i]), "tokens_per_iter": tok / it,
                            "tokens_per_forward": tok / fw})
        return per_seq_out, metrics

    @torch.inference_mode()
    def generate_rollout_records(self, seq: Sequence, n_token_seq_len: Optional[int] = None, return_metrics: bool = False):
        out = self.generate_rollout_records_batch([seq], n_token_seq_len=n_token_seq_len, return_metrics=return_metrics)
        if not return_metrics:
            return out[0] if out else {}
        records, metrics = out
        return (records[0] if records else {}), (metrics[0] if metrics else {})
