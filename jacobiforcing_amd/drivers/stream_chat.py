"""Streaming chat over the multiblock Jacobi loop — the call surface of the reference's
``applications/jacobi_streaming_driver.py:7-193`` (``jacobi_stream_chat``): same arguments, same return tuple
``(assistant_text, final_token_ids, total_new_tokens_est, gen_time)``, ``on_text`` called with the growing text per token
or per accepted chunk.  ``model`` is anything carrying a ``jf_backend`` (see ``hf_seam.Qwen2Backend``); the generation calls
go through ``hf_seam.jacobi_forward_greedy_multiblock`` (HIP loop body), so the reference's driver itself also runs
unmodified once that function is patched onto its model class.

Random draft tails come from ``torch.randint`` exactly where the reference draws them (lines 64-70, 97-104)."""
from __future__ import annotations

import time
from typing import Callable, Dict, List, Optional

import torch

from ..hf_seam import jacobi_forward_greedy_multiblock

MAX_CALLS = 128                                     # jacobi_streaming_driver.py:49


@torch.inference_mode()
def jacobi_stream_chat(model, tokenizer, messages: List[Dict[str, str]], n_token_seq_len: int = 64, max_new_tokens: int = 512,
                       K: int = 2, r: float = 0.8, n_gram_pool_size: int = 4, on_text: Optional[Callable[[str], None]] = None,
                       stream_per_token: bool = True):
    eos_id, pad_id = tokenizer.eos_token_id, tokenizer.pad_token_id
    prompt = tokenizer.apply_chat_template(messages, tokenize=False, add_generation_prompt=True)
    dev = model.jf_backend.device
    input_ids = tokenizer([prompt], return_tensors="pt")["input_ids"].to(dev)
    prompt_len = input_ids.shape[1]
    kw = dict(n_token_seq_len=n_token_seq_len, K=K, r=r, n_gram_pool_size=n_gram_pool_size, tokenizer=tokenizer,
              eos_token_id=eos_id, pad_token_id=pad_id, use_cache=True)
    generated_ids = input_ids.clone()
    total_new, calls, text, jacobi_time = 0, 0, "", 0.0
    cache = first_correct = ngram = None
    while True:
        part = generated_ids[0, prompt_len:]
        if eos_id is not None and part.numel() > 0 and bool((part == eos_id).any()):
            break
        if total_new >= max_new_tokens or calls >= MAX_CALLS:
            break
        if cache is None:                                                        # prefill with a random draft (lines 62-92)
            idxs = torch.randint(low=0, high=generated_ids.shape[1], size=(n_token_seq_len,), device=dev)
            cache, first_correct, ngram, _ = jacobi_forward_greedy_multiblock(
                model, input_ids=torch.cat((input_ids, generated_ids[0, idxs].unsqueeze(0)), dim=-1),
                attention_mask=torch.ones_like(input_ids), past_key_values=None, prefill_phase=True, **kw)
            calls += 1
            continue
        if calls == 1:
            draft = ngram
        else:                                                                    # lines 96-108
            idxs = torch.randint(low=0, high=generated_ids.shape[1], size=(max(n_token_seq_len - 1, 1),), device=dev)
            draft = torch.cat((first_correct.view(1, -1), generated_ids[0, idxs].unsqueeze(0)), dim=-1)
        t0 = time.perf_counter()
        cache, first_correct, accepted, _ = jacobi_forward_greedy_multiblock(
            model, input_ids=draft, attention_mask=None, past_key_values=cache, prefill_phase=False, **kw)
        jacobi_time += time.perf_counter() - t0
        calls += 1
        if accepted is None or accepted.numel() == 0:
            continue
        generated_ids = torch.cat((generated_ids, accepted.to(dev)), dim=-1)
        token_ids = [t for t in accepted[0].tolist() if pad_id is None or t != pad_id]
        eos_hit, usable = False, []
        for t in token_ids:
            if eos_id is not None and t == eos_id:
                eos_hit = True
                break
            usable.append(t)
        if usable:
            total_new += len(usable)
            pieces = [[t] for t in usable] if stream_per_token else [usable]
            for piece in pieces:
                delta = tokenizer.decode(piece, skip_special_tokens=True, clean_up_tokenization_spaces=False)
                if not delta:
                    continue
                text += delta
                if on_text is not None:
                    on_text(text)
        if eos_hit:
            break
    assistant_text = text.strip()
    final_token_ids = torch.empty((1, 0), dtype=torch.long)
    if assistant_text:
        final_token_ids = tokenizer(assistant_text, add_special_tokens=False, return_tensors="pt").input_ids
    return assistant_text, final_token_ids, total_new, jacobi_time
