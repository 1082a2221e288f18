"""Probe (not a test): one prove_batch call over the 300 receipts of a block-shaped trie (BASELINE config 4): rate, and the
chip heights the proofs were made with (how many shapes the grouping left).  ZKSP_TRACE_BATCH=1 adds the timeline."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
zk = importlib.import_module("zk-state-proofs_amd")
mpt = importlib.import_module("zk-state-proofs_amd.mpt")
client = zk.ProverClient(device=0, max_batch=192)
pk, vk = client.setup(zk.merkle_elf())
receipts = mpt.synthetic_block_receipts(300, seed=12)
trie = mpt.block_trie(receipts)
bufs = [mpt.block_proof_input(trie, i).to_borsh() for i in range(300)]
for rep in range(2):
    stdins = []
    for b in bufs:
        s = zk.SP1Stdin(); s.write(b); stdins.append(s)
    t0 = time.perf_counter()
    proofs, status = client.prove_batch(pk, stdins)
    el = time.perf_counter() - t0
    assert status == [0] * 300
    print(f"call {rep}: 300 receipts in {el:.3f} s = {300 / el:.1f} proofs/s", file=sys.stderr, flush=True)
    if rep == 1:
        import collections
        hs = collections.Counter(p.to_bytes()[8:8 + 4 * zk.MACHINE_CHIPS] for p in proofs)
        for k, v in sorted(hs.items(), key=lambda kv: -kv[1]):
            print(v, [int.from_bytes(k[4*c:4*c+4], "little") for c in range(zk.MACHINE_CHIPS)], file=sys.stderr)
    del proofs
