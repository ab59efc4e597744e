"""Checkpoint interchange with the reference trainer (reference: nerf/utils.py:938-1059) and the
nn.Linear <-> FFMLP weight-blob conversion.

File layout written by Trainer.save_checkpoint: a dict {epoch, global_step, stats, [mean_count, mean_density],
[optimizer, lr_scheduler, scaler, ema], model = state_dict}.  The module tree of this package registers the same
parameter / buffer names as the reference (tests/golden/state_dict_keys.json pins that), so a reference `.pth` loads
with load_state_dict.  Files are read with `weights_only=True`: nothing in the file is executed.
"""
import glob
import math
import os

import torch


def _inert_numpy_globals():
    """The pickled globals of a numpy SCALAR: Trainer.save_checkpoint stores `stats` with PSNRMeter.measure() values (V / N with V a
    numpy.float64, nerf/utils.py:212,923,980), which the weights-only unpickler rejects by default.  These three rebuild a scalar
    from raw bytes and execute nothing from the file."""
    import numpy as np
    try:
        from numpy._core.multiarray import scalar          # numpy >= 2
    except ImportError:                                     # numpy 1.x
        from numpy.core.multiarray import scalar
    return [scalar, np.dtype] + sorted({type(np.dtype(t)) for t in ("float64", "float32", "int64", "int32", "bool")}, key=repr)


def _read(path, map_location="cpu"):
    """weights-only load (nothing in the file is executed) with the inert numpy-scalar globals allow-listed"""
    with torch.serialization.safe_globals(_inert_numpy_globals()):
        return torch.load(path, map_location=map_location, weights_only=True)


def latest_checkpoint(ckpt_path, name="ngp"):
    """nerf/utils.py:1001-1008: newest `<name>_ep*.pth` in the workspace's checkpoint directory, or None."""
    files = sorted(glob.glob(os.path.join(ckpt_path, f"{name}_ep*.pth")))
    return files[-1] if files else None


def load_checkpoint(model, checkpoint, map_location=None):
    """Model-only load (load_checkpoint(model_only=True), nerf/utils.py:1010-1033).

    Accepts a full trainer dict or a bare state dict.  Returns (missing_keys, unexpected_keys, meta) with meta holding
    epoch / global_step / stats when present.  `best` checkpoints drop `density_grid` (:987-988): it is reported as missing
    and the bitfield, which is kept, stays usable for rendering."""
    if map_location is None:
        try:
            map_location = next(model.parameters()).device
        except StopIteration:
            map_location = "cpu"
    ckpt = _read(checkpoint, map_location)
    if "model" not in ckpt:
        res = model.load_state_dict(ckpt)
        return list(res.missing_keys), list(res.unexpected_keys), {}
    res = model.load_state_dict(ckpt["model"], strict=False)
    if getattr(model, "cuda_ray", False):
        if "mean_count" in ckpt:
            model.mean_count = ckpt["mean_count"]
        if "mean_density" in ckpt:
            model.mean_density = ckpt["mean_density"]
    meta = {k: ckpt[k] for k in ("epoch", "global_step", "stats") if k in ckpt}
    return list(res.missing_keys), list(res.unexpected_keys), meta


def save_checkpoint(model, path, epoch=0, global_step=0, stats=None, best=False, extra=None):
    """Write the reference's dict layout (nerf/utils.py:943-976).  `extra` may carry optimizer / lr_scheduler / scaler / ema
    state dicts (the `full=True` keys)."""
    state = {"epoch": epoch, "global_step": global_step,
             "stats": stats if stats is not None else {"loss": [], "valid_loss": [], "results": [], "checkpoints": [], "best_result": None}}
    if getattr(model, "cuda_ray", False):
        state["mean_count"] = model.mean_count
        state["mean_density"] = model.mean_density
    if extra:
        state.update(extra)
    sd = model.state_dict()
    if best and "density_grid" in sd:
        sd = {k: v for k, v in sd.items() if k != "density_grid"}
    state["model"] = sd
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(state, path)
    return path


# ---------------------------------------------------------------- nn.Linear <-> FFMLP
def linear_weights_to_ffmlp_blob(weights, hidden_dim=None):
    """[W_0 (h,in), W_1 (h,h), ..., W_n (out,h)] (nn.Linear.weight, bias-free) -> (blob, input_dim, output_dim, num_layers).

    The FFMLP blob is [hidden x in_pad | (num_layers-1) x hidden x hidden | 16 x hidden] row-major (ffmlp.cu:631-634) with
    n+1 matrices for num_layers = n.  Input columns are zero-padded to a multiple of 16 (ffmlp.py:118) and the
    output rows to 16 (ffmlp.py:121); zero padding leaves the function unchanged when the padded inputs are fed zeros."""
    if len(weights) < 3:
        raise ValueError("FFMLP needs at least 3 matrices (num_layers >= 2)")
    ws = [w.detach().float().cpu() for w in weights]
    h = ws[0].shape[0] if hidden_dim is None else hidden_dim
    if h not in (16, 32, 64, 128, 256):
        raise ValueError(f"FFMLP only support hidden_dim in [16, 32, 64, 128, 256], but got {h}")
    in_dim, out_dim = ws[0].shape[1], ws[-1].shape[0]
    if out_dim > 16:
        raise ValueError(f"FFMLP current only supports output dim <= 16, but got {out_dim}")
    for k, w in enumerate(ws):
        want = (h, in_dim) if k == 0 else (out_dim, h) if k == len(ws) - 1 else (h, h)
        if tuple(w.shape) != want:
            raise ValueError(f"layer {k}: expected weight of shape {want}, got {tuple(w.shape)}")
    in_pad = int(math.ceil(in_dim / 16)) * 16
    first = torch.zeros(h, in_pad)
    first[:, :in_dim] = ws[0]
    last = torch.zeros(16, h)
    last[:out_dim] = ws[-1]
    blob = torch.cat([first.reshape(-1)] + [w.reshape(-1) for w in ws[1:-1]] + [last.reshape(-1)])
    return blob, in_pad, out_dim, len(ws) - 1


def ffmlp_blob_to_linear_weights(blob, input_dim, output_dim, hidden_dim, num_layers):
    """Inverse of the above: the n+1 nn.Linear weight matrices of an FFMLP(num_layers=n) blob (padded output rows dropped)."""
    blob = blob.detach().float().cpu().reshape(-1)
    want = hidden_dim * (input_dim + hidden_dim * (num_layers - 1) + 16)
    if blob.numel() != want:
        raise ValueError(f"blob has {blob.numel()} elements, expected {want}")
    out, off = [], 0
    out.append(blob[off:off + hidden_dim * input_dim].view(hidden_dim, input_dim).clone())
    off += hidden_dim * input_dim
    for _ in range(num_layers - 1):
        out.append(blob[off:off + hidden_dim * hidden_dim].view(hidden_dim, hidden_dim).clone())
        off += hidden_dim * hidden_dim
    out.append(blob[off:].view(16, hidden_dim)[:output_dim].clone())
    return out
