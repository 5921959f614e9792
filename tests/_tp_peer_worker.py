"""One rank of a tensor-parallel group whose ranks are SEPARATE PROCESSES sharing GPU 0 (tests/test_tp_peer_gpu.py starts
tp of these). The shards are linked by the peer-to-peer exchange only (RCCL refuses two ranks on one device): export /
gather / import / self-check / agree over files, then the sharded decode step runs as one captured graph per process.
usage: _tp_peer_worker.py RANK TP SHAPE MIX WORKDIR"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, tp_size, shape, mix, workdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    from _pkg import load_package
    pkg = load_package()
    from tinyllama_cpp_amd import engine as E
    from tinyllama_cpp_amd import tp
    E.init(0)
    cfg, tensors = pkg.synth.build_model(shape, mix, seed=5)
    m = E.Model(cfg, tp_rank=rank, tp_size=tp_size)
    rv = tp.FileRendezvous(workdir, rank, tp_size)
    linked = tp.peer_link(m, rv.gather, rv.agree)
    if not linked:
        print(f"rank {rank}: peer link refused: {E.lib().tl_last_error().decode(errors='replace')}")
        return 3
    mode = sys.argv[6] if len(sys.argv) > 6 else ""
    if mode in ("rank1-late", "rank1-late-barrier"):
        # rank 1 arrives 1.5 s late at finalize, whose warm-up step is a collective of the group. Without a host barrier
        # rank 0 waits in vain and tl_model_finalize must SAY so (error 5), not leave a stale flag behind; with the
        # barrier (what bench.py and tools/peer_latency.py do) everything is as if nobody had been late.
        import time
        for name, (typ, arr, rows, cols) in tensors.items():
            m.set_tensor(name, typ, arr, rows, cols)
        if rank == 1:
            time.sleep(1.5)
        if mode == "rank1-late-barrier":
            rv.agree(True)
        outcome, logits = "ok", np.zeros(1, np.float32)
        try:
            m.finalize()
            logits = m.decode(3, 0)[0]
        except E.capi.TlError as e:
            outcome = str(e)
        np.savez(os.path.join(workdir, f"result_{rank}.npz"), linked=True, outcome=outcome, logits=logits)
        rv.agree(True)
        m.free()
        return 0
    m.load(tensors)
    if mode == "rank1-absent":
        # rank 1 stays linked and mapped but never steps: rank 0's exchange must give up at its deadline and the call
        # must return an error (no hang), after which both ranks leave together
        outcome = "n/a"
        if rank == 0:
            try:
                m.decode(1, 0)
                outcome = "no error"
            except E.capi.TlError as e:
                outcome = str(e)
        rv.gather(outcome.encode())
        np.savez(os.path.join(workdir, f"result_{rank}.npz"), linked=True, outcome=outcome)
        rv.agree(True)
        m.free()
        return 0
    toks = np.random.default_rng(2).integers(0, cfg["vocab"], 9).astype(np.int32)
    logits = np.stack([m.decode(int(t), pos)[0] for pos, t in enumerate(toks)])
    nxt = int(np.argmax(logits[-1]))
    ids = m.generate_greedy(nxt, len(toks), 24)
    m.reset_kv()
    ptoks = np.random.default_rng(3).integers(0, cfg["vocab"], 40).astype(np.int32)
    last = m.prefill(ptoks, 0)  # no communicator: token by token through the linked decode graph
    np.savez(os.path.join(workdir, f"result_{rank}.npz"), linked=True, logits=logits, ids=ids, last=last,
             captured=m.graph_captured(), peer=m.tp_peer_enabled(), fused=m.tp_peer_fused(), nodes=m.graph_nodes())
    rv.agree(True)  # nobody unmaps its area while another rank may still be writing into it
    m.free()
    return 0


if __name__ == "__main__":
    sys.exit(main())
