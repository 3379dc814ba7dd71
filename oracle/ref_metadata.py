"""Oracle (test infrastructure): the per-step input/metadata builder, in plain Python lists.

Restates ModelRunner._prepare_inputs / _prepare_block_tables / get_slot_mapping,
nanovllm/engine/model_runner.py:39-148, without the device copies: integers only.
A sequence is any object with the reference's Sequence fields (prompt_tokens, output_tokens,
num_computed_tokens, num_scheduled_tokens, block_table, is_prefill, num_tokens).
"""
from __future__ import annotations


def slot_mapping_ref(block_ids, start_pos: int, end_pos: int, block_size: int):
    """model_runner.py:137-146."""
    return [block_ids[p // block_size] * block_size + p % block_size for p in range(start_pos, end_pos)]


def prepare_inputs_ref(seqs, block_size: int):
    """model_runner.py:46-119 -> dict of python lists / ints."""
    input_ids, positions, slot_mapping = [], [], []
    cu_q, cu_k = [0], [0]
    max_q = max_k = 0
    for seq in seqs:
        if seq.is_prefill:                                           # :59-70
            start = seq.num_computed_tokens
            end = start + seq.num_scheduled_tokens
            input_ids.extend(seq.prompt_tokens[start:end])
            positions.extend(range(start, end))
            slots = slot_mapping_ref(seq.block_table, start, end, block_size)
        else:                                                        # :71-81
            input_ids.append(seq.output_tokens[-1])
            positions.append(seq.num_tokens - 1)
            cur = seq.num_computed_tokens
            slots = slot_mapping_ref(seq.block_table, cur, cur + 1, block_size)
        slot_mapping.extend(slots)
        n_q = seq.num_scheduled_tokens                               # :85-91
        n_k = seq.num_computed_tokens + seq.num_scheduled_tokens
        cu_q.append(cu_q[-1] + n_q)
        cu_k.append(cu_k[-1] + n_k)
        max_q = max(max_q, n_q)
        max_k = max(max_k, n_k)
    width = max(len(s.block_table) for s in seqs)                    # :117-118
    block_tables = [list(s.block_table) + [-1] * (width - len(s.block_table)) for s in seqs]
    return dict(input_ids=input_ids, positions=positions, slot_mapping=slot_mapping,
                cu_seqlens_q=cu_q, cu_seqlens_k=cu_k, max_seqlen_q=max_q, max_seqlen_k=max_k,
                block_tables=block_tables)
