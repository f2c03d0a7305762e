#!/usr/bin/env python3
"""How far is the sampling paths' DEFINITION of the bf16 target distribution (exact softmax of the bf16 scaled logits, rounded
once: include/jacobiforcing.h "Non-greedy verify", oracle exact_softmax_rows) from what the reference would execute on this
machine — ``torch.softmax(logits / T, dim=-1)`` on torch-ROCm bf16 tensors (JDN:64-70, no .float() anywhere) — and how often
could a decision see the difference?  VERDICT r05 Weak #1 / Next #2a.

Per logits regime, at V = 152 064 over >= 10^7 entries:
  * entries whose bf16 value differs, by how many bf16 ulps (bit-pattern distance);
  * accept tests (u < p[draft], u a 24-bit uniform, JDN:328-340): the fraction of uniforms that decide differently, averaged
    over (a) a draft id drawn from p itself — what a converging Jacobi draft looks like — (b) the row's mode, (c) a uniform id;
  * residual draws (inverse CDF over the rounded row, JDN:135-146): the measure of uniforms that land on a different id
    (1 - total overlap of the two normalised CDF partitions);
  * jf_rs_probs itself on the same rows: every p_draft (lower candidate + undecided bit) against the definition, entry by entry
    (one library call per vocabulary id, all rows at once), and how often torch's value is neither candidate.

    python tools/softmax_band.py [--rows 66] [--skip-library] > profiles/softmax_band_r06.txt
"""
import argparse
import ctypes as C
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from jacobiforcing_amd import _native as N  # noqa: E402
from jacobiforcing_amd import ops  # noqa: E402

V = 152064


def bf16_bits_rne(x64: torch.Tensor) -> torch.Tensor:
    """float64 (>= 0) -> bf16 bit patterns (int32) with ONE round-to-nearest-even (no double rounding through float32)."""
    f32 = x64.float()
    b = f32.view(torch.int32)
    hi, low = b >> 16, b & 0xFFFF
    up = (low > 0x8000) | ((low == 0x8000) & ((hi & 1) == 1))          # plain RNE of the float32
    tie = low == 0x8000                                               # float32 sits on a bf16 tie: was the float64 really there?
    back = f32.double()
    up = torch.where(tie & (back > x64), torch.zeros_like(up), up)    # float32 had been rounded UP onto the tie: below it
    up = torch.where(tie & (back < x64), torch.ones_like(up), up)     # rounded DOWN onto the tie: above it
    return hi + up.to(torch.int32)


def bits_to_f64(bits: torch.Tensor) -> torch.Tensor:
    return (bits << 16).view(torch.float32).double()


def regimes(rows: int, dev):
    g = torch.Generator(device="cpu").manual_seed(1234)
    out = []
    for name, std, peak in (("flat  N(0, 0.3)           (random-init weights: the bench's model)", 0.3, None),
                            ("broad N(0, 2)", 2.0, None),
                            ("peaked N(0, 2) + one id at +14 (0.9-0.99 of the mass: a trained checkpoint's confident rows)", 2.0, 14.0),
                            ("peaked N(0, 3) + one id at +9  (~0.2-0.6 of the mass)", 3.0, 9.0)):
        x = torch.randn(rows, V, generator=g) * std
        if peak is not None:
            ids = torch.randint(0, V, (rows,), generator=g)
            x[torch.arange(rows), ids] = peak + torch.rand(rows, generator=g) * 2
        out.append((name, x.to(torch.bfloat16).to(dev)))
    return out


def measure(name, logits, T, use_library: bool):
    dev = logits.device
    R = logits.shape[0]
    xs = logits / float(T) if T != 1.0 else logits                     # JDN:67-68 (bf16 tensor / python float -> bf16)
    ref = torch.softmax(xs, dim=-1)                                    # what the reference runs on this GPU
    exact = torch.softmax(xs.double(), dim=-1)
    dbits = bf16_bits_rne(exact)
    rbits = ref.view(torch.int16).to(torch.int32) & 0xFFFF
    diff = (rbits - dbits)
    n = diff.numel()
    nd = int((diff != 0).sum())
    print(f"## {name}, T = {T}: {R} rows x {V} = {n} entries")
    print(f"   entries whose bf16 value differs from the definition: {nd} = {100.0 * nd / n:.4f} %")
    for k in (-2, -1, 1, 2):
        c = int((diff == k).sum())
        if c or abs(k) == 1:
            print(f"      torch - definition = {k:+d} ulp: {c}")
    big = int((diff.abs() > 2).sum())
    print(f"      |difference| > 2 ulp: {big}")
    p_ref, p_def = bits_to_f64(rbits), bits_to_f64(dbits)
    # accept tests: number of 24-bit uniforms k / 2^24 with k / 2^24 < p is ceil(p 2^24)
    S = float(1 << 24)
    flips = (torch.ceil(p_ref * S) - torch.ceil(p_def * S)).abs() / S   # per entry: fraction of uniforms deciding differently
    w = exact                                                           # draft id ~ p
    mode = exact.argmax(dim=-1)
    ar = torch.arange(R, device=dev)
    print(f"   accept test u < p[draft] decided differently, fraction of 24-bit uniforms:")
    print(f"      draft id drawn from p : {float((flips * w).sum(-1).mean()):.3e}  (max over rows {float((flips * w).sum(-1).max()):.3e})")
    print(f"      draft id = the mode   : {float(flips[ar, mode].mean()):.3e}  (rows whose mode entry differs: {int((diff[ar, mode] != 0).sum())} of {R})")
    print(f"      draft id uniform      : {float(flips.mean()):.3e}")
    # residual draws: partitions of [0, 1) by the normalised running sums of the two rounded rows
    c_ref, c_def = torch.cumsum(p_ref, -1), torch.cumsum(p_def, -1)
    t_ref, t_def = c_ref[:, -1:], c_def[:, -1:]
    b1, b2 = c_ref / t_ref, c_def / t_def
    a1, a2 = (c_ref - p_ref) / t_ref, (c_def - p_def) / t_def
    overlap = (torch.minimum(b1, b2) - torch.maximum(a1, a2)).clamp_(min=0).sum(-1)
    moved = (1.0 - overlap).clamp_(min=0)
    print(f"   inverse-CDF draw landing on a different id, fraction of uniforms: mean {float(moved.mean()):.3e}, max over rows {float(moved.max()):.3e}")
    print(f"      (row sums of the rounded probabilities: torch {float(t_ref.min()):.6f}..{float(t_ref.max()):.6f}, definition {float(t_def.min()):.6f}..{float(t_def.max()):.6f})")
    if not use_library:
        return
    # jf_rs_probs on the same rows: p_draft for EVERY entry (one call per vocabulary id, all rows at once)
    lib = N.lib()
    flat = logits.contiguous()
    f32 = lambda k: torch.zeros((k,), dtype=torch.float32, device=dev)
    p_draft, row_max, row_sumexp = f32(R), f32(R), f32(R)
    packed = ops.new_packed(R, dev)
    ws = torch.zeros((int(lib.jf_rs_workspace_bytes(R, V)) // 4 + 4,), dtype=torch.float32, device=dev)
    ids = torch.zeros((R,), dtype=torch.int64, device=dev)
    got = torch.empty((V, R), dtype=torch.int32, device=dev)
    stream = ops._stream(dev)
    t0 = time.perf_counter()
    for v in range(V):
        ids.fill_(v)
        packed.zero_()
        rc = lib.jf_rs_probs(ops._ptr(flat), N.JF_BF16, R, V, V, ops._ptr(ids), float(T), ops._ptr(p_draft), ops._ptr(row_max),
                             ops._ptr(row_sumexp), ops._ptr(packed), ops._ptr(ws), ws.numel() * 4, stream)
        if rc:
            N.check(rc, "jf_rs_probs")
        got[v].copy_(p_draft.view(torch.int32))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    got = got.t().contiguous()                                          # [R, V] float32 bit patterns of p_draft
    flag = got < 0                                                      # sign bit: the float32 sum could not decide the rounding
    lower = ((got & 0x7FFFFFFF) >> 16)                                  # bf16 bits of |p_draft| (a float holding a bf16 value)
    not_bf16 = int(((got & 0xFFFF) != 0).sum())
    okd = (dbits == lower) | (flag & (dbits == lower + 1))
    okr = (rbits == lower) | (flag & (rbits == lower + 1))
    print(f"   jf_rs_probs, every entry ({V} calls x {R} rows, {dt:.1f} s): p_draft that are not bf16 values: {not_bf16}")
    print(f"      entries flagged undecided (two candidates): {int(flag.sum())} = {100.0 * int(flag.sum()) / n:.3f} %")
    print(f"      definition is NOT the candidate / one of the two candidates: {int((~okd).sum())}   <- must be 0")
    print(f"      torch-ROCm's value is neither candidate: {int((~okr).sum())} = {100.0 * int((~okr).sum()) / n:.4f} %")
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=66)
    ap.add_argument("--skip-library", action="store_true")
    ap.add_argument("--library-regimes", type=int, default=2, help="how many regimes also sweep jf_rs_probs over every entry (T = 0.8)")
    ap.add_argument("--device", default="cuda:0", help="cpu: a dry run of the torch-only part (torch-CPU's softmax, not the reference's device)")
    a = ap.parse_args()
    dev = torch.device(a.device)
    where = torch.cuda.get_device_name(0) if dev.type == "cuda" else "CPU (dry run)"
    print(f"# tools/softmax_band.py: torch {torch.__version__} on {where}, bf16 logits [{a.rows}, {V}]")
    print("# definition = exp(xs - max) / sum in float64 rounded ONCE to bf16, xs = bf16(logits / T) as torch computes it;")
    print("# torch = torch.softmax(logits / T, dim=-1) on the bf16 tensor (the reference's _softmax_with_temperature, JDN:64-70)")
    for i, (name, x) in enumerate(regimes(a.rows, dev)):
        for T in (0.8, 1.0):
            measure(name, x, T, use_library=(not a.skip_library) and dev.type == "cuda" and T == 0.8 and i in (0, 2)[:a.library_regimes])
            print()


if __name__ == "__main__":
    main()
