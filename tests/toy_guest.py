"""Hand-assembled RV32IM guests for tests: a few dozen instructions in a one-segment ELF, with the SP1 syscall epilogue the
verifier needs (sixteen COMMIT / COMMIT_DEFERRED_PROOFS calls carrying sha256 of the empty public values and a zero deferred
digest, then HALT(0)).  No toolchain: the encodings below are the RISC-V unprivileged spec's R / I / U formats."""
import hashlib
import struct

TEXT = 0x00200000
T0, A0, A1 = 5, 10, 11
M32 = 0xFFFFFFFF


def r_type(f7, rs2, rs1, f3, rd, opc=0x33):
    return (f7 << 25) | (rs2 << 20) | (rs1 << 15) | (f3 << 12) | (rd << 7) | opc


def i_type(imm, rs1, f3, rd, opc=0x13):
    return ((imm & 0xFFF) << 20) | (rs1 << 15) | (f3 << 12) | (rd << 7) | opc


def lui(rd, imm20):
    return ((imm20 & 0xFFFFF) << 12) | (rd << 7) | 0x37


def li(rd, v):
    """lui + addi: any 32-bit constant (two instructions, always)."""
    v &= M32
    hi = ((v + 0x800) >> 12) & 0xFFFFF
    return [lui(rd, hi), i_type(v & 0xFFF, rd, 0, rd)]


ECALL = 0x73
MULDIV = {"mul": 0, "mulh": 1, "mulhsu": 2, "mulhu": 3, "div": 4, "divu": 5, "rem": 6, "remu": 7}


def muldiv(name, rd, rs1, rs2):
    return r_type(1, rs2, rs1, MULDIV[name], rd)


def epilogue():
    """COMMIT the digest of the (empty) public values word by word, a zero deferred digest, HALT(0)."""
    dg = struct.unpack("<8I", hashlib.sha256(b"").digest())
    out = []
    for code, words in ((0x10, dg), (0x1A, (0,) * 8)):
        for i, w in enumerate(words):
            out += li(A0, i) + li(A1, w) + li(T0, code) + [ECALL]
    out += li(A0, 0) + li(T0, 0) + [ECALL]
    return out


def elf_of(instrs, symbols=None):
    """ELF32 little-endian RISC-V executable: one PT_LOAD (R+X) segment holding `instrs` at TEXT, entry at its start.
    symbols: {name: address} of FUNC symbols (a .symtab / .strtab pair behind the code), or none."""
    code = b"".join(struct.pack("<I", w & M32) for w in instrs)
    ehsize, phsize = 52, 32
    off = ehsize + phsize
    tail, shoff, shnum = b"", 0, 0
    if symbols:
        strtab = b"\0"
        symtab = bytes(16)  # the null symbol
        for name, addr in symbols.items():
            symtab += struct.pack("<IIIBBH", len(strtab), addr, 4, 0x12, 0, 1)  # GLOBAL FUNC
            strtab += name.encode() + b"\0"
        sym_off = off + len(code)
        str_off = sym_off + len(symtab)
        shoff = str_off + len(strtab)
        sh = bytes(40)  # the null section
        sh += struct.pack("<IIIIIIIIII", 0, 2, 0, 0, sym_off, len(symtab), 2, 1, 4, 16)   # .symtab, linked to section 2
        sh += struct.pack("<IIIIIIIIII", 0, 3, 0, 0, str_off, len(strtab), 0, 0, 1, 0)    # .strtab
        tail, shnum = symtab + strtab + sh, 3
    eh = b"\x7fELF" + bytes([1, 1, 1, 0]) + bytes(8)
    eh += struct.pack("<HHIIIIIHHHHHH", 2, 0xF3, 1, TEXT, ehsize, shoff, 0, ehsize, phsize, 1, 40, shnum, 0)
    ph = struct.pack("<IIIIIIII", 1, off, TEXT, TEXT, len(code), len(code), 5, 4)
    return eh + ph + code + tail


def semantics(name, b, c):
    """RV32IM result of a multiply / divide instruction on 32-bit register values (the spec's corner cases included)."""
    sb = b - (1 << 32) if b >> 31 else b
    sc = c - (1 << 32) if c >> 31 else c
    if name == "mul":
        return (b * c) & M32
    if name == "mulhu":
        return (b * c) >> 32
    if name == "mulh":
        return ((sb * sc) >> 32) & M32
    if name == "mulhsu":
        return ((sb * c) >> 32) & M32
    if name in ("divu", "remu"):
        if c == 0:
            return M32 if name == "divu" else b
        return b // c if name == "divu" else b % c
    if c == 0:
        return M32 if name == "div" else b
    if sb == -(1 << 31) and sc == -1:
        return b if name == "div" else 0
    q = abs(sb) // abs(sc)
    if (sb < 0) != (sc < 0):
        q = -q
    r = sb - q * sc
    return (q if name == "div" else r) & M32
