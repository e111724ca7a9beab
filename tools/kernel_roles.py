"""One table of kernel-symbol patterns for the measurement tools: bench.py (which launch a `roofline.traffic` figure belongs to) and
tools/roofline_table.py (which family a traced dispatch belongs to) read the SAME strings, and tests/test_host_cpu.py checks every
first-choice pattern against the symbols of the built library -- a renamed kernel or a changed template parameter list fails a CPU test
instead of silently turning `roofline.traffic` into null.

Patterns are substrings of the MANGLED symbol (what rocprofv3 --pmc reports in Kernel_Name): the kernel name with its leading `I` (start of
the template argument list) and the run of `Li<n>E` / `Lb<n>E` integer / bool arguments that selects the instantiation."""

# kernel families, longest / most specific first (a dispatch belongs to the first family whose name occurs in its symbol)
FAMILIES = ("gemm8p_kernel", "gemm8_kernel", "gemm_kernel", "gemv_rows_norm_loop_kernel", "gemv_rows_norm_kernel", "gemv_rows_longk_kernel",
            "gemv_rows_kernel", "gemv_xs_split_kernel", "gemv_xs_kernel", "gemv_pk_kernel", "gemv_kernel", "attn_decode_dma_kernel",
            "attn_decode_multi_kernel", "attn_decode_kv8_walk_kernel", "attn_decode_kernel", "attn_merge_mid_kernel", "attn_merge_kernel", "attn2_kernel", "attn_kernel", "vit_qknorm_kernel",
            "vit_qk_sumsq_kernel", "vit_knorm_slots_kernel", "row_sumsq_kernel", "stats_finish_kernel", "fold_cols_kernel", "resid16_norm_kernel", "resid_rmsnorm_kernel", "rmsnorm_kernel", "layernorm_kernel",
            "rope_kv_kernel", "gather_rows_kernel", "argmax_stage1_kernel", "argmax_stage2_kernel", "im2col_kernel", "vit_assemble_kernel",
            "copy_rows_kernel")

EPI_SWIGLU = 4


def decode_gate_up_b1():
    """batch-1 decode gate|up launch (the bench line's dominant kernel): the default form first, then the forms behind tuning keys 38 / 16 / 14.
    gemv_rows_norm_kernel<T, EPI, RR (pairs per wave), NCH (K chunks per lane: 3584 / 512), F8, WAVES>"""
    return [["gemv_rows_norm_kernelI", f"Li{EPI_SWIGLU}ELi1ELi7ELb0E"],
            ["gemv_rows_norm_loop_kernelI", f"Li{EPI_SWIGLU}ELi7ELb0E"],
            ["gemv_rows_norm_kernelI", f"Li{EPI_SWIGLU}ELi3ELi7ELb0E"],
            ["gemv_rows_kernelI", f"Li{EPI_SWIGLU}ELi4ELi4E"]]


def decode_gate_up_batched(b):
    """2 <= b <= 32: gemv_xs_kernel<T, EPI, NB (16-row batch blocks), ...>"""
    return [["gemv_xs_kernelI", f"Li{EPI_SWIGLU}ELi{2 if b > 16 else 1}E"]]


def prefill_gate_up():
    """gemm8_kernel<T, EPI, F8>"""
    return [["gemm8_kernelI", f"Li{EPI_SWIGLU}ELb0E"]]


def first_choice_patterns():
    """(role, substrings) pairs that MUST match a symbol of the built library"""
    return [("decode gate|up, batch 1", decode_gate_up_b1()[0]), ("decode gate|up, batch 32", decode_gate_up_batched(32)[0]),
            ("decode gate|up, batch 16", decode_gate_up_batched(16)[0]), ("prefill gate|up", prefill_gate_up()[0])]
