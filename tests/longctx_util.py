"""Helpers of the long-context attention tests: the oracle evaluated for single (query row, head) pairs,
optionally with one logical cache block taken out of the sequence, so that a test can show that its own
bound would catch a kernel that lost a block of keys (tests/conftest.py: assert_bound_is_sensitive)."""
import torch

from oracle import ref_attention

_ONE = torch.tensor([0, 1], dtype=torch.int32)


def head_slices(k_cache, v_cache, hkv):
    """CPU copies of one kv head of the caches, [NB, bs, 1, D]."""
    return k_cache[:, :, hkv:hkv + 1].cpu(), v_cache[:, :, hkv:hkv + 1].cpu()


def top_key_block(q_vec, kc1, bt_row, n_k):
    """Logical block index of the key with the largest score for this query vector."""
    k = ref_attention.gather_kv(kc1, bt_row, n_k)[:, 0].float()
    return int((k @ q_vec.float()).argmax()) // kc1.shape[1]


def oracle_head_row(q_vec, kc1, vc1, bt_row, n_k, scale, drop=None):
    """Oracle output [D] of one query vector that sees keys 0..n_k-1 of one kv head (kc1, vc1: that head's
    slices); ``drop`` removes logical block ``drop`` (a full one) from the sequence first."""
    bs = kc1.shape[1]
    bt_row = bt_row.cpu().to(torch.int32)
    if drop is not None:
        assert (drop + 1) * bs <= n_k, "only whole blocks can be dropped"
        bt_row = torch.cat([bt_row[:drop], bt_row[drop + 1:]])
        n_k -= bs
    return ref_attention.paged_attention_ref(q_vec.cpu()[None, None, :], kc1, vc1, _ONE,
                                             torch.tensor([0, n_k], dtype=torch.int32), bt_row[None], scale)[0, 0]


def check_rows_with_sensitivity(out_of, samples, q, k_cache, v_cache, G, scale, name, peaked, assert_close, assert_sensitive):
    """samples: list of (token row t, query head h, block-table row, n_k).  ``out_of(t, h)`` gives the
    kernel's [D] output.  Every sample is compared under the scale-aware bound; every sample with at least
    two whole blocks also proves the bound's sensitivity (drop a block: a pseudo-random one for diffuse
    softmax, the one holding the largest score for peaked softmax, where most blocks carry no weight)."""
    worst, least = 0.0, float("inf")
    by_head = {}
    for s in samples:
        by_head.setdefault(s[1] // G, []).append(s)
    for hkv, group in sorted(by_head.items()):
        kc1, vc1 = head_slices(k_cache, v_cache, hkv)
        bs = kc1.shape[1]
        for t, h, bt_row, n_k in group:
            qv = q[t, h].cpu()
            ref = oracle_head_row(qv, kc1, vc1, bt_row, n_k, scale)
            worst = max(worst, assert_close(out_of(t, h), ref, f"{name} row {t} head {h} n_k {n_k}"))
            full = n_k // bs
            if full >= 2:
                drop = top_key_block(qv, kc1, bt_row.cpu(), full * bs) if peaked else (7 * t + h) % full
                damaged = oracle_head_row(qv, kc1, vc1, bt_row, n_k, scale, drop=drop)
                least = min(least, assert_sensitive(ref, damaged, f"{name} row {t} head {h} without block {drop}"))
    return worst, least
