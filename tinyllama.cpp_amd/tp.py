"""Tensor-parallel sharding of the model's tensors (host side of SURVEY §8e).

Megatron-style: attn_q/k/v, ffn_gate/up and output.weight are split by OUTPUT ROWS (rank r holds the r-th block of
query heads / kv heads / intermediate columns / vocab rows); attn_output and ffn_down are split by INPUT COLUMNS
(the r-th block, which must be a whole number of 256-element super-blocks); token_embd and the F32 norm vectors are
replicated. Per layer there are two sum all-reduces of f32[hidden] (after attn_output and after ffn_down; rank 0
folds the residual into its partial) and one all-gather of the logits at the end. This replaces the reference's
request-level MPI routing (distributed_coordinator.cpp), which never touched tensors.

Pure numpy on raw GGUF block arrays ([nblocks, block_bytes] uint8, row-major by output row): usable on the CPU
(tests/test_tp_gloo.py runs the sharded forward over gloo) and by the GPU engine (engine.Model(tp_rank, tp_size)).
"""
import numpy as np

from .synth import BLOCK_BYTES, BLOCK_ELEMS, F32

ROW_SPLIT = ("attn_q", "attn_k", "attn_v", "ffn_gate", "ffn_up")
COL_SPLIT = ("attn_output", "ffn_down")


def check(cfg, size):
    """Raise ValueError unless `cfg` shards `size` ways (heads, kv heads, 256-column boundaries)."""
    H, I, nh, nkv, V = cfg["hidden"], cfg["inter"], cfg["n_heads"], cfg["n_kv_heads"], cfg["vocab"]
    hd = H // nh
    if size < 1 or nh % size or nkv % size or I % size or V % size:
        raise ValueError(f"tp_size {size} does not divide heads {nh} / kv heads {nkv} / intermediate {I} / vocab {V}")
    if size > 1 and ((nh // size * hd) % 256 or (I // size) % 256):
        raise ValueError(f"tp_size {size}: column shards of attn_output ({nh // size * hd}) / ffn_down ({I // size}) "
                         "are not multiples of 256")


def kind_of(name):
    """'row' | 'col' | 'rep' for a GGUF tensor name."""
    if name == "output.weight":
        return "row"
    if name.startswith("blk."):
        k = name.split(".")[2]
        if k in ROW_SPLIT:
            return "row"
        if k in COL_SPLIT:
            return "col"
    return "rep"


def shard_tensor(name, typ, arr, rows, cols, rank, size):
    """(arr_local, rows_local, cols_local) of rank `rank` out of `size` for one full tensor."""
    kind = kind_of(name)
    if size == 1 or kind == "rep":
        return arr, rows, cols
    if typ == F32:
        a = np.asarray(arr, np.float32).reshape(rows, cols)
        if kind == "row":
            r = rows // size
            return np.ascontiguousarray(a[rank * r:(rank + 1) * r]), r, cols
        c = cols // size
        return np.ascontiguousarray(a[:, rank * c:(rank + 1) * c]), rows, c
    be, bb = BLOCK_ELEMS[typ], BLOCK_BYTES[typ]
    a = np.asarray(arr, np.uint8).reshape(rows, cols // be, bb)
    if kind == "row":
        assert rows % size == 0, (name, rows, size)
        r = rows // size
        return np.ascontiguousarray(a[rank * r:(rank + 1) * r]).reshape(-1, bb), r, cols
    assert cols % size == 0 and (cols // size) % be == 0, (name, cols, size)
    c = cols // size
    return np.ascontiguousarray(a[:, rank * c // be:(rank + 1) * c // be]).reshape(-1, bb), rows, c


def local_dims(cfg, size):
    """Per-rank dimensions: query heads, kv heads, intermediate columns, vocab rows."""
    return dict(n_heads=cfg["n_heads"] // size, n_kv_heads=cfg["n_kv_heads"] // size, inter=cfg["inter"] // size,
                vocab=cfg["vocab"] // size)


# ---- wiring of the peer-to-peer exchange (csrc/tl_peer.hip) ---------------------------------------------------------------
def peer_link(model, gather, agree):
    """Link the shards of a group by the one-shot peer exchange, or leave them on the library collectives.

    model: engine.Model of this rank (tp_size 2..8, not finalized). gather(b: bytes) -> [bytes of rank 0, 1, ...];
    agree(flag: bool) -> logical AND of `flag` over all ranks. Both are collective calls of whatever channel the host
    already has (torch.distributed in bench.py, files in the one-GPU test). Returns True when every rank passed the
    self-check and the exchange was enabled on all of them; on False nothing changed (RCCL stays in charge)."""
    handle, good = b"\0" * 64, True
    try:
        handle = model.tp_peer_export()
    except Exception:
        good = False
    handles = gather(handle)
    if not agree(good):
        return False
    try:
        good = model.tp_peer_import(handles)
    except Exception:
        good = False
    good = agree(good)
    if good:
        model.tp_peer_enable(True)
    return good


class FileRendezvous:
    """gather / agree for peer_link over files in a directory shared by the ranks (processes of one host)."""

    def __init__(self, directory, rank, world, timeout=120.0):
        import os
        self.dir, self.rank, self.world, self.timeout, self.seq = directory, rank, world, timeout, 0
        os.makedirs(directory, exist_ok=True)

    def gather(self, payload):
        import os
        import time
        self.seq += 1
        mine = os.path.join(self.dir, f"x{self.seq}_{self.rank}")
        with open(mine + ".tmp", "wb") as f:
            f.write(payload)
        os.replace(mine + ".tmp", mine)  # atomic: a file that exists is complete
        out, t0 = [], time.time()
        for r in range(self.world):
            path = os.path.join(self.dir, f"x{self.seq}_{r}")
            while not os.path.exists(path):
                if time.time() - t0 > self.timeout:
                    raise TimeoutError(f"rank {r} did not show up at rendezvous step {self.seq}")
                time.sleep(0.01)
            with open(path, "rb") as f:
                out.append(f.read())
        return out

    def agree(self, flag):
        return all(b == b"1" for b in self.gather(b"1" if flag else b"0"))
