"""Executor acceptance: the committed reference guest
(fixtures/sp1-merkle-proof.elf = reference circuits/elf/riscv32im-succinct-zkvm-elf,
sha256 b58c1c11...ae53079) must behave exactly as measured in SURVEY.md appendix
A.4 / A.5: cycle counts, memory-op counts, keccak-f call counts, syscall census,
opcode histogram, public values, sha256 digest words, and the panic path."""
import hashlib

import numpy as np
import pytest

GOLDEN = [  # (name, fixture call, node sizes, cycles, keccak calls, memory ops)
    ("acct-d1", lambda fx: fx.acct_fixture(1), [108], 105208, 3, 35278),
    ("tx-d2", lambda fx: fx.tx_fixture(), [532, 116], 359949, 14, 126932),
    ("acct-d4", lambda fx: fx.acct_fixture(4), [532] * 3 + [106], 710670, 30, 251500),
    ("acct-d6", lambda fx: fx.acct_fixture(6), [532] * 5 + [105], 1052890, 46, 373727),
    ("acct-d7", lambda fx: fx.acct_fixture(7), [532] * 6 + [105], 1230192, 54, 436088),
    ("acct-d8", lambda fx: fx.acct_fixture(8), [532] * 7 + [104], 1406960, 62, 498303),
    ("acct-d9", lambda fx: fx.acct_fixture(9), [532] * 8 + [104], 1578131, 70, 559438),
    ("acct-d10", lambda fx: fx.acct_fixture(10), [532] * 9 + [103], 1749075, 78, 620437),
]

A5_PINS = {  # depth -> (root prefix, root suffix, borsh len, sha256(borsh))
    1: ("de99fe12", "1497055f", 188, "021b08a3d564c24ab314f402579702b140300b0723a5c97393f43a29ba5190b7"),
    2: ("fce34689", "454a0245", 723, "fc2fc25318927ccde152bfb74684ac6387451680d74a149c432489aa24372722"),
    4: ("2e7fe759", "3c43e711", 1794, "4a49b994445dc0902495b15a2b21ae4eb006258d2fe946e706c58236aa7cccbc"),
    5: ("970a3c13", "f464d470", 2330, "27a999b014da00801f10bdb5dcb956c1054405a425a34acd5ab84b372fd50c27"),
    7: ("bc860208", "51c4e455", 3401, "b61ac465a144fbaa749d8a40d5724dbd7e9b95447a9957dddaf86769afbe1832"),
    8: ("f018e309", "75dc698e", 3936, "9ff514eed4deade2737815d1171bda7b020928d357f19f0229d3301090cd10f7"),
}

ACCT_D8_HIST = dict(xor=241060, lw=236255, sw=196607, addi=107901, slli=99669, srli=94728, xori=74695, lbu=29956,
                    sb=24435, bne=24210, add=19511, beq=15139, lb=11033, jalr=10190, andi=9342, bltu=7475, lui=5909,
                    auipc=5457, sltiu=3087, sltu=2878, sub=2150, bgeu=2136, jal=1953, blt=1496, mul=359, ori=179,
                    sll=104, srai=47, mulhu=30, ecall=21, sh=17, srl=12, bge=3)
ACCT_D8_HIST["or"] = 100973
ACCT_D8_HIST["and"] = 77943


def test_elf_fixture_is_the_reference_binary(zk):
    assert hashlib.sha256(zk.merkle_elf()).hexdigest() == \
        "b58c1c119c713e48b45dca052fa8c600a6bc47a2deb190784777787c1ae53079"


def test_fixture_generator_pins(fx):
    assert fx.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert fx.keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    assert fx.keccak256(b"\x80").hex() == "56e81f171bcc55a6ff8345e692c0f86e5b48e01b996cadc001622fb5e363b421"
    assert fx.keccak256(fx.USDT_ADDRESS).hex() == "ab14d68802a763f7db875346d03fbf86f137de55814b191c069e721f47474733"
    for d, (pre, suf, blen, digest) in A5_PINS.items():
        m = fx.acct_fixture(d)
        b = m.to_borsh()
        assert m.root_hash.hex().startswith(pre) and m.root_hash.hex().endswith(suf)
        assert len(b) == blen and hashlib.sha256(b).hexdigest() == digest
        assert fx.MerkleProofInput.from_borsh(b) == m


def test_rlp_receipt_known_answer(fx):
    """The one hermetic vector of the reference (trie-utils/tests/rlp.rs:12):
    rlp([status=false, cumGas=1, 256-byte zero bloom, [log(addr, 2 topics, data)]])."""
    log = fx.rlp_list([fx.rlp_bytes(bytes(19) + b"\x11"),
                       fx.rlp_list([fx.rlp_bytes(bytes(30) + b"\xde\xad"), fx.rlp_bytes(bytes(30) + b"\xbe\xef")]),
                       fx.rlp_bytes(bytes.fromhex("0100ff"))])
    got = fx.rlp_list([fx.rlp_uint(0), fx.rlp_uint(1), fx.rlp_bytes(bytes(256)), fx.rlp_list([log])])
    expected = ("f901668001b90100" + "00" * 256 + "f85ff85d94" + "00" * 19 + "11" + "f842a0" + "00" * 30 + "dead"
                + "a0" + "00" * 30 + "beef" + "830100ff")
    assert got.hex() == expected and len(got) == 361


@pytest.mark.parametrize("name,make,sizes,cycles,nkeccak,memops", GOLDEN, ids=[g[0] for g in GOLDEN])
def test_golden_runs(zk, fx, host_client, oracle, name, make, sizes, cycles, nkeccak, memops):
    pk, vk = host_client.setup(zk.merkle_elf())
    m = make(fx)
    assert [len(n) for n in m.proof] == sizes
    s = zk.SP1Stdin()
    s.write(m.to_borsh())
    rep, pv, err, rc = host_client.execute(pk, s, zk.KECCAK_OBSERVE)
    assert rc == 0 and rep.exit_code == 0
    assert (rep.cycles, rep.n_keccak, rep.memory_ops) == (cycles, nkeccak, memops)
    assert list(rep.syscalls) == [1, 2, 8, 8, 1, 1]  # HALT, WRITE, COMMIT, COMMIT_DEFERRED, HINT_LEN, HINT_READ
    # public values = the leaf the reference's verify_merkle_proof returns, digest = sha256 of them
    assert pv == oracle.verify_merkle_proof(m.root_hash, m.proof, m.key)
    assert np.array(list(rep.pv_digest), dtype="<u4").tobytes() == hashlib.sha256(pv).digest()
    # permutation count rule of SURVEY.md appendix A.4
    per = lambda n: -(-(n + 1) // 136)
    assert nkeccak == 2 * sum(per(len(n)) for n in m.proof) + per(len(m.proof[0]))
    # software mode runs the same instruction stream; replace mode keeps the answers
    rep0, pv0, _, rc0 = host_client.execute(pk, s, zk.KECCAK_SOFTWARE)
    assert rc0 == 0 and rep0.cycles == cycles and rep0.n_keccak == 0 and pv0 == pv
    rep2, pv2, _, rc2 = host_client.execute(pk, s, zk.KECCAK_REPLACE)
    assert rc2 == 0 and rep2.n_keccak == nkeccak and pv2 == pv and rep2.cycles < cycles
    assert list(rep2.pv_digest) == list(rep.pv_digest)


def test_acct_d8_details(zk, fx, host_client):
    pk, _ = host_client.setup(zk.merkle_elf())
    s = zk.SP1Stdin()
    s.write(fx.acct_fixture(8).to_borsh())
    rep, pv, err, rc = host_client.execute(pk, s, zk.KECCAK_OBSERVE)
    assert pv == fx.ACCOUNT_VALUE and len(pv) == 70
    assert np.array(list(rep.pv_digest), dtype="<u4").tobytes().hex() == \
        "002da47712e44be9ce52a7048da1e48f0f047a9cf7aeefee347c2030f40a6246"
    hist = host_client.opcode_histogram(rep)
    assert hist == ACCT_D8_HIST and sum(hist.values()) == 1406960
    # the first keccak-f input is node[0][0..136] absorbed into a zero state
    st = host_client.keccak_states(pk, s)
    assert st.shape == (62, 25)
    node0 = fx.acct_fixture(8).proof[0]
    assert st[0].tobytes() == node0[:136] + bytes(64)


def test_other_fixture_families(zk, fx, host_client):
    pk, _ = host_client.setup(zk.merkle_elf())
    for make, cycles, nk in ((lambda: fx.slot_fixture(0), 875273, 38), (lambda: fx.receipt_fixture(5), 759107, 32)):
        s = zk.SP1Stdin()
        s.write(make().to_borsh())
        rep, pv, err, rc = host_client.execute(pk, s, zk.KECCAK_OBSERVE)
        assert rc == 0 and rep.exit_code == 0 and rep.n_keccak == nk
        assert abs(int(rep.cycles) - cycles) < 2000  # generator details (seed, cumGas) differ from the survey's scratch run


def test_tampered_proof_panics_like_the_reference(zk, fx, host_client):
    pk, _ = host_client.setup(zk.merkle_elf())
    m = fx.acct_fixture(8)
    node = bytearray(m.proof[3])
    node[-1] ^= 1
    m.proof[3] = bytes(node)
    s = zk.SP1Stdin()
    s.write(m.to_borsh())
    rep, pv, err, rc = host_client.execute(pk, s, zk.KECCAK_OBSERVE)
    assert rc == 0 and rep.exit_code == 1 and rep.cycles == 1314554 and pv == b""
    assert "panicked at crypto-ops-0.1.0/src/lib.rs:21:10" in err
    assert "Failed to verify Merkle Proof: InvalidProof" in err


def test_malformed_inputs(zk, host_client):
    pk, _ = host_client.setup(zk.merkle_elf())
    s = zk.SP1Stdin()
    s.write(b"\x01\x02\x03")  # not a borsh MerkleProofInput: guest unwrap() panics
    rep, pv, err, rc = host_client.execute(pk, s, zk.KECCAK_OBSERVE)
    assert rc == 0 and rep.exit_code != 0 and pv == b""
    empty = zk.SP1Stdin()  # no stdin entry at all: executor-level fault, not a guest panic
    rep, pv, err, rc = host_client.execute(pk, empty, zk.KECCAK_OBSERVE)
    assert rc != 0 and "input stream exhausted" in host_client.last_error()
    with pytest.raises(zk.ZkspError):
        host_client.setup(b"not an elf")


def test_stale_json_guest_is_a_second_regression_input(zk, fx, host_client):
    """The reference also commits an older build of the same guest that takes serde_json input
    (reference elf/riscv32im-succinct-zkvm-elf, the source shown in README.md:101-116; SURVEY.md
    section 2 row 4): 2 203 254 cycles on the depth-8 account fixture, same public values."""
    import json
    import os
    path = os.path.join(os.path.dirname(zk.MERKLE_ELF_PATH), "sp1-merkle-proof-json.elf")
    elf = open(path, "rb").read()
    assert hashlib.sha256(elf).hexdigest() == "0d9935d51e05c09e6e0f89d6055c62f493c82608b0f850138e73dbd27469d5c1"
    pk, _ = host_client.setup(elf)
    m = fx.acct_fixture(8)
    js = json.dumps({"proof": [list(n) for n in m.proof], "root_hash": list(m.root_hash), "key": list(m.key)},
                    separators=(",", ":")).encode()
    s = zk.SP1Stdin()
    s.write(js)
    rep, pv, _, rc = host_client.execute(pk, s, zk.KECCAK_OBSERVE)
    assert rc == 0 and rep.exit_code == 0
    assert rep.cycles == 2203254
    assert pv == fx.ACCOUNT_VALUE
    assert bytes(np.array(list(rep.pv_digest), np.uint32).tobytes()) == hashlib.sha256(pv).digest()
    # software and observed runs execute the same instruction stream
    rep0, _, _, _ = host_client.execute(pk, s, zk.KECCAK_SOFTWARE)
    assert rep0.cycles == rep.cycles


def test_precompile_key_is_refused_for_a_function_that_is_not_keccak_f(zk, host_client):
    """The precompile shape replaces calls of the functions an ELF names keccakf by the keccak chip, so its verifying key
    stands for "those functions are keccak-f".  Key generation checks it: every such function is run by the executor on test
    states and compared with keccak-f[1600] (executor.hpp check_keccakf_entries) - the committed guest's two pass (every
    other test's setup), a function of that name that does something else is refused.  The as-committed shape, which
    replaces nothing, takes any ELF."""
    import toy_guest as tg
    ret = tg.i_type(0, 1, 0, 0, opc=0x67)  # jalr x0, 0(ra)
    body = tg.li(tg.A0, 0x00300000) + [0, 0] + tg.epilogue()  # (two slots for the call, patched below)
    fn = tg.TEXT + 4 * len(body)
    call = 2  # index of the first slot
    # jal ra, fn
    off = fn - (tg.TEXT + 4 * call)
    jal = (((off >> 20) & 1) << 31) | (((off >> 1) & 0x3FF) << 21) | (((off >> 11) & 1) << 20) | (((off >> 12) & 0xFF) << 12) | (1 << 7) | 0x6F
    body[call] = jal
    body[call + 1] = tg.i_type(0, 0, 0, 0)  # nop
    for name, fbody in (("tiny_keccak::keccakf::keccakf", [ret]),                                      # returns at once: the state is unchanged
                        ("keccakf", [tg.i_type(1, tg.A0, 0, tg.A0), tg.r_type(0, 0, 0, 0, 0), ret])):   # clobbers a0, still not keccak-f
        elf = tg.elf_of(body + fbody, symbols={name: fn})
        with pytest.raises(zk.ZkspError) as ei:
            host_client.setup(elf)
        assert "does not compute keccak-f" in str(ei.value)
        # a function that never returns properly is refused too
    loop = tg.elf_of(body + [0x0000006F], symbols={"keccakf": fn})  # jal x0, 0: spins until the cycle limit
    with pytest.raises(zk.ZkspError) as ei:
        host_client.setup(loop)
    assert "did not return" in str(ei.value)
    # without the symbol the same program is an ordinary guest; with it, the as-committed shape (keccak_mode 1) replaces nothing
    host_client.setup(tg.elf_of(body + [ret]))
    zk.ProverClient(device=-1, keccak_mode=1).setup(tg.elf_of(body + [ret], symbols={"keccakf": fn}))
