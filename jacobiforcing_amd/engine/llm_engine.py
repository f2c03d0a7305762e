"""``LLMEngine`` — request plumbing with the reference's surface (inference_engine/engine/llm_engine.py:15-202):
``add_request``, ``step``, ``is_finished``, ``generate`` returning ``[{"text", "token_ids"}]``, ``exit``."""
from __future__ import annotations

import atexit
import sys
from dataclasses import fields
from time import perf_counter

from ..config import Config
from ..sampling_params import SamplingParams
from .model_runner import ModelRunner
from .scheduler import Scheduler
from .sequence import Sequence, SequenceStatus


class LLMEngine:
    def __init__(self, model, tokenizer_path=None, **kwargs):
        config_fields = {f.name for f in fields(Config)}
        config = Config(model, **{k: v for k, v in kwargs.items() if k in config_fields})       # ENG:20-22
        self.config = config
        self.ps, self.events, self._exited = [], [], False
        self.model_runner = ModelRunner(config, 0, self.events, device=kwargs.get("device"))
        self.tokenizer = None
        tok_path = tokenizer_path or model
        if tok_path != "none":
            try:
                from transformers import AutoTokenizer
                self.tokenizer = AutoTokenizer.from_pretrained(tok_path, use_fast=True)
            except Exception as e:  # no tokenizer files: token-id prompts still work
                print(f"[LLMEngine] tokenizer not loaded from {tok_path} ({type(e).__name__}); prompts must be token ids", file=sys.stderr, flush=True)
        if self.tokenizer is not None:
            config.eos = self.tokenizer.eos_token_id
            config.pad = self.tokenizer.pad_token_id if self.tokenizer.pad_token_id is not None else self.tokenizer.eos_token_id
        else:
            config.eos, config.pad = config.hf_config.eos_token_id, config.hf_config.pad_token_id
        self.scheduler = Scheduler(config)
        self.model_runner.block_manager = self.scheduler.block_manager
        self.scheduler.set_kv_cache(self.model_runner.kv_cache)
        self.scheduler.release_row = self.model_runner.release
        atexit.register(self.exit)

    def exit(self):
        if getattr(self, "_exited", False):
            return
        self._exited = True
        if getattr(self, "model_runner", None) is not None:
            self.model_runner.call("exit")
            self.model_runner = None

    def add_request(self, prompt, sampling_params: SamplingParams):
        if isinstance(prompt, str):
            if self.tokenizer is None:
                raise ValueError("string prompts need a tokenizer; pass token ids or a tokenizer_path")
            prompt = self.tokenizer.encode(prompt)
        seq = Sequence(prompt, sampling_params)
        self.scheduler.add(seq)

    def step(self):
        seqs, is_prefill = self.scheduler.schedule()
        result = self.model_runner.call("run", seqs, is_prefill)
        if result is None or (is_prefill and len(result) > 0 and result[0] == []):              # ENG:90-93
            return [], sum(len(seq) for seq in seqs)
        if len(result) > 0 and isinstance(result[0], dict):                                     # on-policy, ENG:95-116
            # every sequence of the batch finishes and carries the WHOLE list of per-sequence records, as in the reference
            num_new = 0 if is_prefill else sum(seq.num_completion_tokens for seq in seqs)
            for seq in seqs:
                seq.status = SequenceStatus.FINISHED
                seq._rollout_records = result
                if seq in self.scheduler.running:
                    self.scheduler.running.remove(seq)
                self.model_runner.release(seq)
            return [(seq.seq_id, seq._rollout_records) for seq in seqs], (-num_new if num_new > 0 else 0)
        token_ids = result
        is_jacobi = len(token_ids) > 0 and isinstance(token_ids[0], (list, tuple))
        if is_jacobi:
            self.scheduler.postprocess_jacobi(seqs, token_ids)
            num_new = sum(len(t) for t in token_ids)
        else:
            self.scheduler.postprocess(seqs, token_ids)
            num_new = len(seqs)
        outputs = []
        for seq in seqs:
            if seq.is_finished:
                outputs.append((seq.seq_id, seq.completion_token_ids))
                self.model_runner.release(seq)
        return outputs, (sum(len(seq) for seq in seqs) if is_prefill else -num_new)

    def is_finished(self):
        return self.scheduler.is_finished()

    def generate(self, prompts, sampling_params, use_tqdm: bool = True):
        pbar = None
        if use_tqdm:
            from tqdm.auto import tqdm
            pbar = tqdm(total=len(prompts), desc="Generating", dynamic_ncols=True)
        if not isinstance(sampling_params, list):
            sampling_params = [sampling_params] * len(prompts)
        for prompt, sp in zip(prompts, sampling_params):
            self.add_request(prompt, sp)
        outputs = {}
        prefill_tp = decode_tp = 0.0
        while not self.is_finished():
            t = perf_counter()
            output, num_tokens = self.step()
            if pbar is not None:
                dt = max(perf_counter() - t, 1e-9)
                if num_tokens > 0:
                    prefill_tp = num_tokens / dt
                else:
                    decode_tp = -num_tokens / dt
                pbar.set_postfix({"Prefill": f"{int(prefill_tp)}tok/s", "Decode": f"{int(decode_tp)}tok/s"})
            for seq_id, toks in output:
                outputs[seq_id] = toks
                if pbar is not None:
                    pbar.update(1)
        if pbar is not None:
            pbar.close()
        ordered = [outputs[seq_id] for seq_id in sorted(outputs)]
        if ordered and isinstance(ordered[0], list) and ordered[0] and isinstance(ordered[0][0], dict):   # ENG:176-185
            records = []
            for seq_output in ordered:          # one copy of the batch's record list per sequence (reference behaviour)
                records.extend(seq_output) if isinstance(seq_output, list) else records.append(seq_output)
            return records
        res = []
        for seq_id in sorted(outputs):
            toks = [int(t) for t in outputs[seq_id]]
            text = self.tokenizer.decode(toks) if self.tokenizer is not None else ""
            res.append({"text": text, "token_ids": toks})
        return res
