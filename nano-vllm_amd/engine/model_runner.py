"""Builds the flat per-step inputs and the CacheContext, owns the KV cache, runs the model, samples.

Metadata integers follow the reference's ModelRunner._prepare_inputs / _prepare_block_tables /
get_slot_mapping exactly (nanovllm/engine/model_runner.py:39-148): prefill rows take
prompt[computed : computed+scheduled] at positions computed..; decode rows take the last sampled
token at position num_tokens-1 with the slot of logical position `num_computed` (both are the slice
[computed, computed+scheduled) of prompt + outputs, which also covers the recomputation of a
preempted sequence); slot =
block_table[p // bs] * bs + p % bs; seqlen_q = scheduled, seqlen_k = computed + scheduled; block
tables are right-padded with -1.  They are built with numpy in one pass (the reference loops per
token and round-trips through the device, model_runner.py:140-148).  The cache has the reference's
layout [2, L, num_blocks, block_size, H_kv, D] (:175-184) and its per-layer views are bound to every
module that has `k_cache` and `v_cache` attributes (:186-192); the number of blocks is sized from
free device memory (the formula the reference left commented out at :171) unless given.
"""
from __future__ import annotations

import numpy as np
import torch

from ..utils.cache_context import CacheContext, set_context
from .sequence import Sequence


def _prepare_decode_inputs(seqs: list[Sequence], block_size: int) -> dict:
    """prepare_inputs for a step in which every sequence has exactly one (decode) token: the same
    integers, built from flat python lists and whole-array numpy operations."""
    B = len(seqs)
    width = max(len(s.block_table) for s in seqs)
    flat: list[int] = []
    pad = [-1] * width
    for s in seqs:
        bt = s.block_table
        flat.extend(bt)
        if len(bt) < width:
            flat.extend(pad[len(bt):])
    tables = np.array(flat, dtype=np.int32).reshape(B, width)
    start = np.array([s.num_computed_tokens for s in seqs], dtype=np.int64)
    input_ids = np.array([s.output_tokens[-1] for s in seqs], dtype=np.int64)
    positions = np.array([len(s) - 1 for s in seqs], dtype=np.int64)
    slots = tables[np.arange(B), start // block_size].astype(np.int64) * block_size + start % block_size
    cu_q = np.arange(B + 1, dtype=np.int32)
    cu_k = np.zeros(B + 1, dtype=np.int32)
    np.cumsum(start + 1, out=cu_k[1:])
    return dict(input_ids=input_ids, positions=positions, slot_mapping=slots.astype(np.int32), cu_seqlens_q=cu_q,
                cu_seqlens_k=cu_k, max_seqlen_q=1, max_seqlen_k=int(start.max()) + 1, block_tables=tables,
                num_decode_seqs=B, num_leading_decode=B, max_seqlen_k_decode=int(start.max()) + 1,
                max_seqlen_q_rest=0, max_seqlen_k_rest=0, sum_seqlen_k_decode=int(cu_k[-1]))


def prepare_inputs(seqs: list[Sequence], block_size: int, decode_fast_path: bool = True) -> dict:
    """Pure host part: python ints / numpy arrays for one step (see module docstring)."""
    if decode_fast_path and all(s.num_scheduled_tokens == 1 and s.output_tokens and s.completes_step for s in seqs):
        return _prepare_decode_inputs(seqs, block_size)
    ids, pos, slots = [], [], []
    n_q = np.empty(len(seqs), dtype=np.int64)
    n_k = np.empty(len(seqs), dtype=np.int64)
    width = max(len(s.block_table) for s in seqs)
    tables = np.full((len(seqs), width), -1, dtype=np.int32)
    n_decode = 0
    for i, seq in enumerate(seqs):
        table = np.asarray(seq.block_table, dtype=np.int64)
        tables[i, :len(table)] = table
        # tokens [start, end) of prompt + outputs: a prompt chunk, the last sampled token (decode:
        # start = num_tokens - 1, which is the reference's position and slot for that row), or a
        # chunk of a preempted sequence's recomputation
        start = seq.num_computed_tokens
        end = start + seq.num_scheduled_tokens
        n_prompt = len(seq.prompt_tokens)
        if end <= n_prompt:
            toks = seq.prompt_tokens[start:end]
        elif start >= n_prompt:
            toks = seq.output_tokens[start - n_prompt:end - n_prompt]
        else:
            toks = seq.prompt_tokens[start:] + seq.output_tokens[:end - n_prompt]
        p = np.arange(start, end, dtype=np.int64)
        ids.append(np.asarray(toks, dtype=np.int64))
        pos.append(p)
        slots.append(table[p // block_size] * block_size + p % block_size)
        n_q[i] = seq.num_scheduled_tokens
        n_k[i] = seq.num_computed_tokens + seq.num_scheduled_tokens
        n_decode += int(seq.num_scheduled_tokens == 1)
    cu_q = np.zeros(len(seqs) + 1, dtype=np.int32)
    cu_k = np.zeros(len(seqs) + 1, dtype=np.int32)
    np.cumsum(n_q, out=cu_q[1:])
    np.cumsum(n_k, out=cu_k[1:])
    single = n_q == 1
    lead = len(seqs) if single.all() else int(np.argmin(single))     # leading rows with one query token
    return dict(input_ids=np.concatenate(ids), positions=np.concatenate(pos),
                slot_mapping=np.concatenate(slots).astype(np.int32), cu_seqlens_q=cu_q, cu_seqlens_k=cu_k,
                max_seqlen_q=int(n_q.max()), max_seqlen_k=int(n_k.max()), block_tables=tables,
                num_decode_seqs=n_decode, num_leading_decode=lead,
                max_seqlen_k_decode=int(n_k[:lead].max()) if lead else 0,
                max_seqlen_q_rest=int(n_q[lead:].max()) if lead < len(seqs) else 0,
                max_seqlen_k_rest=int(n_k[lead:].max()) if lead < len(seqs) else 0,
                sum_seqlen_k_decode=int(cu_k[lead]))


class _DecodeGraph:
    """A captured decode step for one (padded batch, padded table width) shape with its static buffers."""

    def __init__(self, runner: "ModelRunner", batch: int, width: int, chunk_keys: int = 0):
        dev = runner.device
        self.batch, self.width, self.chunk_keys = batch, width, chunk_keys
        # All per-step metadata lives in ONE device buffer with a pinned host mirror, so a step is one
        # host-to-device copy: [input_ids i64 x B | positions i64 x B | slot_mapping i32 x B |
        # cu_seqlens_k i32 x (B+1) | block_tables i32 x B*W], every section 8-byte aligned.
        def sec(n_items, itemsize, off):
            nbytes = -(-n_items * itemsize // 8) * 8
            return off, off + n_items * itemsize, off + nbytes
        o_ids = sec(batch, 8, 0)
        o_pos = sec(batch, 8, o_ids[2])
        o_slot = sec(batch, 4, o_pos[2])
        o_cuk = sec(batch + 1, 4, o_slot[2])
        o_bt = sec(batch * width, 4, o_cuk[2])
        o_src = sec(batch, 8, o_bt[2])      # pipelined decode: row of the previous step's tokens, or -1
        self._dev_buf = torch.zeros(o_src[2], dtype=torch.uint8, device=dev)
        # Two pinned host mirrors used alternately, each guarded by an event recorded after its
        # host-to-device copy: with pipelined decode the host runs ahead of the device, and a mirror
        # must not be rewritten while an earlier step's asynchronous copy may still be reading it.
        self._host_bufs = [torch.zeros(o_src[2], dtype=torch.uint8, pin_memory=dev.type == "cuda") for _ in range(2)]
        self._copied = [None, None]
        self._turn = 0

        def views(buf):
            return (buf[o_ids[0]:o_ids[1]].view(torch.int64), buf[o_pos[0]:o_pos[1]].view(torch.int64),
                    buf[o_slot[0]:o_slot[1]].view(torch.int32), buf[o_cuk[0]:o_cuk[1]].view(torch.int32),
                    buf[o_bt[0]:o_bt[1]].view(torch.int32).view(batch, width),
                    buf[o_src[0]:o_src[1]].view(torch.int64))
        (self.input_ids, self.positions, self.slot_mapping, self.cu_seqlens_k, self.block_tables,
         self.src_rows) = views(self._dev_buf)
        self._hs = [[t.numpy() for t in views(buf)] for buf in self._host_bufs]
        for h in self._hs:
            h[5].fill(-1)
            h[2].fill(-1)
            h[4].fill(-1)
        self._dev_buf.copy_(self._host_bufs[0])
        self.cu_seqlens_q = torch.arange(batch + 1, dtype=torch.int32, device=dev)
        self.ctx = CacheContext(layer_idx=-1, is_prefill=False, slot_mapping=self.slot_mapping,
                                cu_seqlens_q=self.cu_seqlens_q, cu_seqlens_k=self.cu_seqlens_k, max_seqlen_q=1,
                                max_seqlen_k=width * runner.block_size, block_tables=self.block_tables,
                                block_size=runner.block_size, num_decode_seqs=batch, num_leading_decode=batch,
                                max_seqlen_k_decode=width * runner.block_size, decode_chunk_keys=chunk_keys)
        set_context(self.ctx)
        # warm-up on a side stream (library initialisation must not happen inside the capture)
        side = runner.capture_stream()          # one side stream for all captures: they share its per-stream buffers
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            runner.model(self.input_ids, self.positions)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        # captured on the stream of the warm-up: per-stream buffers of the launches (split-key workspaces, the
        # zero-initialised one of small decode launches) exist already and are the ones the graph keeps using
        with torch.cuda.graph(self.graph, stream=side):
            self.logits = runner.model(self.input_ids, self.positions)

    def replay(self, m: dict, src_rows: np.ndarray | None = None, prev_tokens: torch.Tensor | None = None) -> torch.Tensor:
        """src_rows / prev_tokens (pipelined decode): row i takes its input token from
        prev_tokens[src_rows[i]] (still on the device) where src_rows[i] >= 0."""
        n = len(m["input_ids"])
        turn = self._turn
        self._turn ^= 1
        if self._copied[turn] is not None:
            self._copied[turn].synchronize()       # the copy that last read this mirror is done
        h_ids, h_pos, h_slot, h_cuk, h_bt, h_src = self._hs[turn]
        h_src[:] = -1
        if src_rows is not None:
            h_src[:n] = src_rows
        h_ids[:n] = m["input_ids"]
        h_pos[:n] = m["positions"]
        h_slot[:n] = m["slot_mapping"]
        h_slot[n:] = -1                            # padded rows: slot -1 (skipped), no keys, token 0
        h_cuk[:n + 1] = m["cu_seqlens_k"]
        h_cuk[n + 1:] = m["cu_seqlens_k"][-1]
        bt = m["block_tables"]
        h_bt[:n, :bt.shape[1]] = bt
        h_bt[:n, bt.shape[1]:] = -1
        h_bt[n:] = -1
        self._dev_buf.copy_(self._host_bufs[turn], non_blocking=True)
        if self._dev_buf.is_cuda:
            if self._copied[turn] is None:
                self._copied[turn] = torch.cuda.Event()
            self._copied[turn].record()
        if src_rows is not None:
            src = self.src_rows[:n]
            self.input_ids[:n] = torch.where(src >= 0, prev_tokens[src.clamp(min=0)], self.input_ids[:n])
        self.graph.replay()
        return self.logits[:n]


class ModelRunner:
    def __init__(self, model: torch.nn.Module, cfg, block_size: int = 256, num_kvcache_blocks: int | None = None,
                 gpu_memory_utilization: float = 0.9, device: str | torch.device = "cuda",
                 dtype: torch.dtype = torch.bfloat16, use_graphs: bool = False):
        self.model, self.cfg, self.block_size = model, cfg, block_size
        self.device, self.dtype = torch.device(device), dtype
        self.kv_cache: torch.Tensor | None = None
        self.num_kvcache_blocks = self.init_kv_cache(num_kvcache_blocks, gpu_memory_utilization)
        # Decode steps (one token per sequence) can be replayed from captured HIP graphs: the
        # kernels of a step are launch-latency-bound on small models.  No counterpart in the
        # reference (its README lists CUDA graphs as future work).
        self.use_graphs = use_graphs and self.device.type == "cuda"
        shard = getattr(model, "shard", None)
        self.sharded = shard is not None and shard.world > 1
        self.group = getattr(model, "group", None)
        if self.use_graphs and self.sharded:
            import torch.distributed as dist
            if dist.get_backend(self.group) != "nccl":
                raise ValueError("use_graphs with a head-sharded model needs the RCCL ('nccl') backend: the per-layer "
                                 f"all-gather of backend {dist.get_backend(self.group)!r} cannot be captured in a HIP graph")
        self._graphs: dict[tuple[int, int, int], _DecodeGraph] = {}
        # plan decode launches with the step's lengths (False: from the longest one alone, as a caller of the
        # reference's interface would; kept as a switch for measurements)
        self.use_length_hints = True

    def capture_stream(self) -> "torch.cuda.Stream":
        if getattr(self, "_capture_stream", None) is None:
            self._capture_stream = torch.cuda.Stream(device=self.device)
        return self._capture_stream

    # ---- KV cache -------------------------------------------------------------------------
    def block_bytes(self) -> int:
        c = self.cfg
        return 2 * c.num_hidden_layers * self.block_size * self.local_kv_heads * c.head_dim * self.dtype.itemsize

    @property
    def local_kv_heads(self) -> int:
        """kv heads whose cache this rank holds (all of them unless the model is head-sharded)."""
        shard = getattr(self.model, "shard", None)
        return shard.local_kv_heads if shard is not None else self.cfg.num_key_value_heads

    def init_kv_cache(self, num_blocks: int | None, utilization: float) -> int:
        c = self.cfg
        if num_blocks is None:
            if self.device.type != "cuda":
                raise ValueError("num_kvcache_blocks must be given when not running on a GPU")
            free, total = torch.cuda.mem_get_info(self.device)
            budget = int(total * utilization) - (total - free)
            num_blocks = budget // self.block_bytes()
        assert num_blocks > 0, "no memory left for the KV cache"
        self.kv_cache = torch.empty(2, c.num_hidden_layers, num_blocks, self.block_size, self.local_kv_heads,
                                    c.head_dim, dtype=self.dtype, device=self.device)
        layer = 0
        for module in self.model.modules():
            if hasattr(module, "k_cache") and hasattr(module, "v_cache"):
                module.k_cache = self.kv_cache[0][layer]
                module.v_cache = self.kv_cache[1][layer]
                layer += 1
        assert layer == c.num_hidden_layers, f"bound {layer} attention layers, expected {c.num_hidden_layers}"
        return num_blocks

    # ---- one step -------------------------------------------------------------------------
    def _to_device(self, arr: np.ndarray) -> torch.Tensor:
        t = torch.from_numpy(arr)
        if self.device.type == "cuda":
            return t.pin_memory().to(self.device, non_blocking=True)
        return t

    def set_step_context(self, seqs: list[Sequence], m: dict | None = None):
        if m is None:
            m = prepare_inputs(seqs, self.block_size)
        ctx = CacheContext(layer_idx=-1, is_prefill=False, slot_mapping=self._to_device(m["slot_mapping"]),
                           cu_seqlens_q=self._to_device(m["cu_seqlens_q"]),
                           cu_seqlens_k=self._to_device(m["cu_seqlens_k"]), max_seqlen_q=m["max_seqlen_q"],
                           max_seqlen_k=m["max_seqlen_k"], block_tables=self._to_device(m["block_tables"]),
                           block_size=self.block_size, num_decode_seqs=m["num_decode_seqs"],
                           num_leading_decode=m["num_leading_decode"], max_seqlen_k_decode=m["max_seqlen_k_decode"],
                           max_seqlen_q_rest=m["max_seqlen_q_rest"], max_seqlen_k_rest=m["max_seqlen_k_rest"],
                           decode_chunk_keys=self._decode_chunk(m, m["max_seqlen_k_decode"]))
        set_context(ctx)
        return self._to_device(m["input_ids"]), self._to_device(m["positions"]), m

    def _decode_chunk(self, m: dict, max_k: int) -> int:
        """Keys per decode work item for this step's decode rows (0 = the launch plan's choice): the runner knows
        every length, the launch plan only `max_k`, so a few long sequences among many short ones are cut finer."""
        nd = m["num_leading_decode"]
        if self.device.type != "cuda" or not nd or not self.use_length_hints:
            return 0
        from .. import ops
        c = self.cfg
        shard = getattr(self.model, "shard", None)
        hq = shard.local_heads if shard is not None else c.num_attention_heads
        return ops.decode_suggest_chunk(nd, hq, self.local_kv_heads, max_k, m["sum_seqlen_k_decode"])

    def _sample(self, logits: torch.Tensor, temperatures: list[float]) -> torch.Tensor:
        tokens = sample(logits, temperatures)
        if self.sharded and any(t != 0 for t in temperatures):
            from ..parallel import broadcast_tokens
            tokens = broadcast_tokens(tokens, self.group)      # rank 0's draw; argmax is identical everywhere
        return tokens

    def _graph_for(self, n_seqs: int, table_width: int, m: dict | None = None) -> _DecodeGraph:
        # batch buckets: 1, 2, 4, 8, then multiples of 16 (the kernels plan their launch from the padded
        # batch, so padding a small batch to 16 would halve its useful waves)
        batch = 1 << max(0, int(n_seqs - 1).bit_length()) if n_seqs <= 8 else -(-n_seqs // 16) * 16
        # table width buckets: powers of two up to 8 blocks, then four per octave (the captured launch
        # plans for width * block_size keys, so a coarse bucket would leave half of its key chunks empty)
        p2 = 1 << max(0, int(table_width - 1).bit_length())
        step = max(1, p2 // 8)
        width = p2 if p2 <= 8 else -(-table_width // step) * step
        # the launch plan is frozen into the capture: a skewed batch (a few long sequences among many short ones)
        # gets a capture of its own, planned with the chunk suggested for it -- rounded up to a power of two or 1.5 x one so that
        # steps of similar skew share one capture (a suggestion is valid for any lengths the width holds)
        chunk = self._decode_chunk(m, width * self.block_size) if m is not None else 0
        if chunk:
            p2 = 1 << int(chunk - 1).bit_length()           # next of 256, 384, 512, 768, 1024, ...
            chunk = p2 * 3 // 4 if chunk <= p2 * 3 // 4 else p2
        g = self._graphs.get((batch, width, chunk))
        if g is None:
            # a captured graph for a slightly larger batch serves (padded rows cost little; a capture
            # costs two eager forwards): only capture when nothing within 25 % exists
            fits = [k for k in self._graphs
                    if k[1] == width and k[2] == chunk and batch <= k[0] <= batch + max(16, batch // 4)]
            if fits:
                g = self._graphs[min(fits)]
            else:
                g = self._graphs[(batch, width, chunk)] = _DecodeGraph(self, batch, width, chunk)
        return g

    @torch.inference_mode()
    def launch_decode(self, seqs: list[Sequence], prev: dict | None) -> dict:
        """Pipelined decode: enqueue a decode-only step without waiting for the previous step's token
        values.  Sequences whose last output token is still PENDING take it on the device from `prev`
        (the handle this method returned for the previous step).  Returns a handle for fetch_tokens()."""
        order = self.decode_order(seqs)
        if order is not None:                               # fetch_tokens() answers in the caller's order
            seqs = [seqs[i] for i in order]
        m = prepare_inputs(seqs, self.block_size)
        src = np.full(len(seqs), -1, dtype=np.int64)
        if prev is not None:
            rows = prev["rows"]
            for i, s in enumerate(seqs):
                if s.num_pending:
                    src[i] = rows[s.id]
        assert not any(s.num_pending and src[i] < 0 for i, s in enumerate(seqs)), "pending token without a source"
        ids = m["input_ids"]
        ids[ids < 0] = 0                                    # placeholders never reach the embedding
        prev_tokens = prev["tokens"] if prev is not None else None
        if self.use_graphs:
            g = self._graph_for(len(seqs), m["block_tables"].shape[1], m)
            set_context(g.ctx)
            logits = g.replay(m, src if prev is not None else None, prev_tokens)
        else:
            input_ids, positions, m = self.set_step_context(seqs, m)
            if prev is not None:
                src_d = self._to_device(src)
                input_ids = torch.where(src_d >= 0, prev_tokens[src_d.clamp(min=0)], input_ids)
            logits = self.model(input_ids, positions)
        tokens = self._sample(logits, [s.sampling_params.temperature for s in seqs])
        handle = {"rows": {s.id: i for i, s in enumerate(seqs)}, "tokens": tokens, "event": None, "order": order}
        if self.device.type == "cuda":
            host = torch.empty(len(seqs), dtype=torch.int64, pin_memory=True)
            host.copy_(tokens, non_blocking=True)
            handle["host"] = host
            handle["event"] = torch.cuda.Event()
            handle["event"].record()
        else:
            handle["host"] = tokens
        return handle

    @staticmethod
    def fetch_tokens(handle: dict) -> list[int]:
        if handle["event"] is not None:
            handle["event"].synchronize()
        tokens = handle["host"].tolist()
        order = handle.get("order")
        if order is not None:
            back = [0] * len(tokens)
            for row, i in enumerate(order):
                back[i] = tokens[row]
            tokens = back
        return tokens

    @torch.inference_mode()
    def run(self, seqs: list[Sequence]) -> list[int | None]:
        """Forward the scheduled tokens; returns one sampled token per sequence, or None while a
        sequence's prompt is still being chunked or its K/V is being recomputed (model_runner.py:214-240)."""
        order = None
        if self.use_graphs and all(s.num_scheduled_tokens == 1 for s in seqs):
            order = self.decode_order(seqs)
            if order is not None:
                seqs = [seqs[i] for i in order]
            m = prepare_inputs(seqs, self.block_size)
            g = self._graph_for(len(seqs), m["block_tables"].shape[1], m)
            set_context(g.ctx)
            logits = g.replay(m)
        else:
            order = self.step_order(seqs)
            if order is not None:
                seqs = [seqs[i] for i in order]
            input_ids, positions, m = self.set_step_context(seqs)
            last_rows = torch.from_numpy(m["cu_seqlens_q"][1:].astype(np.int64) - 1).to(self.device)
            logits = self.model(input_ids, positions, logits_rows=last_rows)      # [B, vocab]
        temps = [s.sampling_params.temperature for s in seqs]
        tokens = self._sample(logits, temps).tolist()
        out = [tok if seq.completes_step else None for seq, tok in zip(seqs, tokens)]
        if order is not None:                       # back to the caller's order
            back = [None] * len(out)
            for row, i in enumerate(order):
                back[i] = out[row]
            out = back
        return out

    sort_decode_rows_from = 64       # decode steps of at least this many rows go to the launch longest first

    @staticmethod
    def decode_order(seqs: list[Sequence]) -> list[int] | None:
        """Row order of a decode-only step: longest sequence first once the step has sort_decode_rows_from rows.
        The decode launch hands out its work items in row order and ends with its last workgroup, so the long
        rows should not be the ones that start last (512 rows of 100 .. 2000 keys: 419 us in random order, 353
        longest first; 256 rows of 513 .. 767 keys: no difference; fewer rows do not fill the chip and have no
        tail to shorten).  None when the order stands."""
        if len(seqs) < ModelRunner.sort_decode_rows_from:
            return None
        lens = np.fromiter((s.num_computed_tokens for s in seqs), dtype=np.int64, count=len(seqs))
        order = np.argsort(-lens, kind="stable")
        if np.array_equal(order, np.arange(len(seqs))):
            return None
        return order.tolist()

    @staticmethod
    def step_order(seqs: list[Sequence]) -> list[int] | None:
        """Row order of a step that holds prompt chunks: the single-token rows first (they are the decode
        kernel's; in the scheduler's order, or longest first as in decode_order()), then the others by key count,
        longest first.  The attention launch
        takes its work in row order and ends when its last workgroup does, so the longest sequences should start
        first (seven 512-token chunks over 512 .. 8192 cached keys: 212 us in scheduler order, 191 longest first).
        None when the order stands."""
        single = [i for i, s in enumerate(seqs) if s.num_scheduled_tokens == 1]
        rest = [i for i, s in enumerate(seqs) if s.num_scheduled_tokens != 1]
        if len(single) >= ModelRunner.sort_decode_rows_from:
            single.sort(key=lambda i: -seqs[i].num_computed_tokens)        # stable: ties keep the scheduler's order
        elif len(rest) < 2 and (not rest or not single or single[-1] < rest[0]):
            return None
        rest.sort(key=lambda i: -(seqs[i].num_computed_tokens + seqs[i].num_scheduled_tokens))
        order = single + rest
        return None if order == list(range(len(seqs))) else order


def sample(logits: torch.Tensor, temperatures: list[float]) -> torch.Tensor:
    """temperature == 0 -> argmax; otherwise softmax(logits / T) + multinomial
    (nanovllm/utils/sampler.py:25-34)."""
    greedy = logits.argmax(dim=-1)
    if all(t == 0 for t in temperatures):
        return greedy
    t = torch.tensor([x if x > 0 else 1.0 for x in temperatures], dtype=torch.float32, device=logits.device)
    probs = torch.softmax(logits.float() / t.unsqueeze(1), dim=-1)
    drawn = torch.multinomial(probs, num_samples=1).squeeze(1)
    is_greedy = torch.tensor([x == 0 for x in temperatures], device=logits.device)
    return torch.where(is_greedy, greedy, drawn)
