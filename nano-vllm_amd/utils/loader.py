"""Weights from a local directory of HF-style ``*.safetensors`` shards into the model by parameter name
(what the reference's nanovllm/utils/loader.py:10-34 does, minus the hub download: there is no network
here, so the path must exist).  With a head-sharded model only this rank's rows of q/k/v_proj are kept."""
from __future__ import annotations

import glob
import os

import torch
from safetensors import safe_open


def load_model(model: torch.nn.Module, model_path: str, strict: bool = True) -> list[str]:
    """Copy every tensor of the checkpoint into the parameter of the same name; returns the names the
    model does not have (the reference prints and skips them, loader.py:29-34).  `strict` additionally
    requires that every model parameter was found (a tied lm_head counts as found)."""
    files = sorted(glob.glob(os.path.join(model_path, "*.safetensors")))
    if not files:
        raise FileNotFoundError(f"no *.safetensors under {model_path}")
    shard = getattr(model, "shard", None)
    d = model.config.head_dim if shard is not None else 0
    params = dict(model.named_parameters())
    seen, unknown = set(), []
    with torch.no_grad():
        for path in files:
            with safe_open(path, framework="pt", device="cpu") as f:
                for name in f.keys():
                    p = params.get(name)
                    if p is None:
                        unknown.append(name)
                        continue
                    t = f.get_tensor(name)
                    if shard is not None and ".self_attn." in name:
                        part = name.rsplit(".", 2)[-2]
                        if part == "q_proj":
                            t = t[shard.q_heads.start * d:shard.q_heads.stop * d]
                        elif part in ("k_proj", "v_proj"):
                            t = t[shard.kv_heads.start * d:shard.kv_heads.stop * d]
                    if t.shape != p.shape:
                        raise ValueError(f"{name}: checkpoint shape {tuple(t.shape)} != parameter {tuple(p.shape)}")
                    p.copy_(t)
                    seen.add(name)
    if strict:
        missing = [n for n in params if n not in seen]
        if missing:
            raise KeyError(f"checkpoint lacks {len(missing)} parameters, e.g. {missing[:3]}")
    return unknown
