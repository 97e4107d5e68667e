"""TensorFlow Saver-V2 checkpoints ("tensor bundles") without TensorFlow: the reference's `saver.save` / `saver.restore`
(BrainDQN.py:176-192,227-228; BrainDQNNature.py:103-115) read and written as plain files, so that weights trained by the
reference load into this framework and checkpoints written here load into the reference.

A bundle `<prefix>` is two files:
    <prefix>.index                an SSTable in LevelDB's table format (uncompressed blocks, prefix-compressed keys): key "" ->
                                  BundleHeaderProto, key <tensor name> -> BundleEntryProto {dtype, shape, shard_id, offset, size, crc32c}
    <prefix>.data-00000-of-00001  the tensors' bytes back to back (little endian), in key order
Pinned by the reference's own files: its train_history/**/bird-*.index (the .data blobs are not shipped with the reference) parse
with this reader into exactly the variable names, shapes and byte layout below; two of them are committed as fixtures
(tests/golden/tf_bundle_*.index, data files of the reference).  The tensor bytes themselves are pinned by a round trip through the writer
(crc32c per tensor and per table block, as TensorFlow checks them).

Variable names (tf.Variable creation order = this framework's flat parameter order, Q1 of SURVEY section 8):
    BrainDQN / BrainPrioritizedReplyDQN-style single net:  Variable, Variable_1 .. Variable_9 (+ /Adam, /Adam_1, beta1_power, beta2_power)
    BrainDQNNature / BrainDoubleDQN:  eval_net/Variable.. , target_net/Variable.. ; the optimizer was created inside the target_net scope:
                                      target_net/eval_net/Variable/Adam .., target_net/beta1_power, target_net/beta2_power
"""
import os
import struct

import numpy as np

MAGIC = bytes.fromhex("57fb808b247547db")       # kTableMagicNumber, little endian
DT_FLOAT = 1
SHAPES = [(8, 8, 4, 32), (32,), (4, 4, 32, 64), (64,), (3, 3, 64, 64), (64,), (1600, 512), (512,), (512, 2), (2,)]      # Variable .. Variable_9


# ------------------------------------------------------------------------------------------------ crc32c (Castagnoli), masked as LevelDB / TF do
def _make_table():
    tab = np.zeros(256, np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
        tab[i] = c
    return tab


_TAB = _make_table()
_TAB8 = None


def crc32c(data, crc=0):
    """CRC-32C of bytes / a uint8 array (slicing-by-8 over numpy for the 3 MB tensors)."""
    global _TAB8
    b = np.frombuffer(bytes(data), np.uint8) if not isinstance(data, np.ndarray) else data.view(np.uint8).ravel()
    if _TAB8 is None:
        t = [_TAB]
        for _ in range(7):
            p = t[-1]
            t.append((p >> 8) ^ _TAB[p & 0xFF])
        _TAB8 = [x.tolist() for x in t]
    c = crc ^ 0xFFFFFFFF
    n8 = len(b) // 8 * 8
    t0, t1, t2, t3, t4, t5, t6, t7 = _TAB8
    if n8:
        words = b[:n8].view("<u4").reshape(-1, 2).tolist()
        for lo, hi in words:
            lo ^= c
            c = (t7[lo & 255] ^ t6[(lo >> 8) & 255] ^ t5[(lo >> 16) & 255] ^ t4[lo >> 24] ^
                 t3[hi & 255] ^ t2[(hi >> 8) & 255] ^ t1[(hi >> 16) & 255] ^ t0[hi >> 24])
    for x in b[n8:].tolist():
        c = t0[(c ^ x) & 255] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask(crc):
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------ varints / protobuf wire format
def _get_varint(b, i):
    r = s = 0
    while True:
        c = b[i]
        i += 1
        r |= (c & 0x7F) << s
        s += 7
        if c < 0x80:
            return r, i


def _put_varint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _proto_fields(b):
    i, out = 0, []
    while i < len(b):
        t, i = _get_varint(b, i)
        fn, wt = t >> 3, t & 7
        if wt == 0:
            v, i = _get_varint(b, i)
        elif wt == 2:
            n, i = _get_varint(b, i)
            v = bytes(b[i:i + n])
            i += n
        elif wt == 5:
            v = struct.unpack_from("<I", b, i)[0]
            i += 4
        elif wt == 1:
            v = struct.unpack_from("<Q", b, i)[0]
            i += 8
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        out.append((fn, v))
    return out


class Entry:
    """BundleEntryProto: where one tensor lives in the data file."""

    def __init__(self, dtype, shape, shard_id, offset, size, crc):
        self.dtype, self.shape, self.shard_id, self.offset, self.size, self.crc32c = dtype, tuple(shape), shard_id, offset, size, crc

    def __repr__(self):
        return f"Entry(dtype={self.dtype}, shape={self.shape}, offset={self.offset}, size={self.size})"

    @staticmethod
    def parse(b):
        f = _proto_fields(b)
        shape = []
        for fn, v in f:
            if fn == 2:
                for fn2, dim in _proto_fields(v):
                    if fn2 == 2:
                        shape.append(dict(_proto_fields(dim)).get(1, 0))
        d = {fn: v for fn, v in f if fn != 2}
        return Entry(d.get(1, 0), shape, d.get(3, 0), d.get(4, 0), d.get(5, 0), d.get(6, 0))

    def serialize(self):
        out = bytearray()
        out += b"\x08" + _put_varint(self.dtype)
        dims = b"".join(b"\x12" + _put_varint(len(x)) + x for x in (b"\x08" + _put_varint(d) for d in self.shape))
        out += b"\x12" + _put_varint(len(dims)) + dims
        if self.shard_id:
            out += b"\x18" + _put_varint(self.shard_id)
        if self.offset:
            out += b"\x20" + _put_varint(self.offset)
        out += b"\x28" + _put_varint(self.size)
        out += b"\x35" + struct.pack("<I", self.crc32c)
        return bytes(out)


# ------------------------------------------------------------------------------------------------ LevelDB table
def _block_entries(blk):
    nrest = struct.unpack_from("<I", blk, len(blk) - 4)[0]
    end = len(blk) - 4 - 4 * nrest
    i, key = 0, b""
    while i < end:
        shared, i = _get_varint(blk, i)
        non_shared, i = _get_varint(blk, i)
        vlen, i = _get_varint(blk, i)
        key = key[:shared] + bytes(blk[i:i + non_shared])
        i += non_shared
        yield key, bytes(blk[i:i + vlen])
        i += vlen


def _read_block(d, off, size, verify):
    body, kind = d[off:off + size], d[off + size]
    if kind != 0:
        raise ValueError("compressed table block (TensorFlow writes bundle indices uncompressed)")
    if verify:
        want = struct.unpack_from("<I", d, off + size + 1)[0]
        if mask(crc32c(d[off:off + size + 1])) != want:
            raise ValueError("table block checksum mismatch")
    return body


def read_index(path, verify=True):
    """<prefix>.index -> (header fields, {tensor name: Entry}) in key order."""
    d = open(path, "rb").read()
    if len(d) < 48 or d[-8:] != MAGIC:
        raise ValueError(f"{path}: not a table file (bad magic)")
    f = d[-48:]
    _, i = _get_varint(f, 0)
    _, i = _get_varint(f, i)                      # metaindex handle
    io, i = _get_varint(f, i)
    isz, i = _get_varint(f, i)
    entries, header = {}, None
    for _, handle in _block_entries(_read_block(d, io, isz, verify)):
        o, j = _get_varint(handle, 0)
        s, j = _get_varint(handle, j)
        for k, v in _block_entries(_read_block(d, o, s, verify)):
            if k == b"":
                header = dict(_proto_fields(v))
            else:
                entries[k.decode()] = Entry.parse(v)
    return header, entries


def _build_block(items, restart_interval=16):
    out, restarts, last = bytearray(), [], b""
    for n, (k, v) in enumerate(items):
        shared = 0
        if n % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(last), len(k)) and last[shared] == k[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def _successor(key):
    """LevelDB BytewiseComparator::FindShortSuccessor: the shortest string >= key (the index key of the last block)"""
    for i, c in enumerate(key):
        if c != 0xFF:
            return key[:i] + bytes([c + 1])
    return key


def _separator(start, limit):
    """FindShortestSeparator: a short string in [start, limit) (the index key between two blocks)"""
    n = 0
    while n < min(len(start), len(limit)) and start[n] == limit[n]:
        n += 1
    if n < min(len(start), len(limit)) and start[n] < 0xFF and start[n] + 1 < limit[n]:
        return start[:n] + bytes([start[n] + 1])
    return start


def build_table(items, block_size=4096):
    """[(key bytes, value bytes)] in key order -> the bytes of a LevelDB table as TensorFlow's table builder lays it out: data blocks of
    ~block_size (restart interval 16), an empty metaindex block, the index block (restart interval 1), the 48-byte footer; every block
    followed by its type byte (0 = uncompressed) and masked crc32c."""
    out = bytearray()

    def emit(body):
        off = len(out)
        out.extend(body + b"\x00")
        out.extend(struct.pack("<I", mask(crc32c(body + b"\x00"))))
        return _put_varint(off) + _put_varint(len(body))

    blocks, cur, size = [], [], 0
    for k, v in items:
        cur.append((k, v))
        size += len(k) + len(v) + 3
        if size >= block_size:
            blocks.append(cur)
            cur, size = [], 0
    if cur:
        blocks.append(cur)
    handles = []
    for n, blk in enumerate(blocks):
        h = emit(_build_block(blk))
        last = blk[-1][0]
        handles.append((_separator(last, blocks[n + 1][0][0]) if n + 1 < len(blocks) else _successor(last), h))
    meta = emit(_build_block([]))
    index = emit(_build_block(handles, restart_interval=1))
    foot = meta + index
    out.extend(foot + b"\x00" * (40 - len(foot)) + MAGIC)
    return bytes(out)


def table_items(path):
    """the raw (key, value) pairs of a table file, in order"""
    d = open(path, "rb").read()
    f = d[-48:]
    _, i = _get_varint(f, 0)
    _, i = _get_varint(f, i)
    io, i = _get_varint(f, i)
    isz, i = _get_varint(f, i)
    items = []
    for _, handle in _block_entries(_read_block(d, io, isz, True)):
        o, j = _get_varint(handle, 0)
        s, j = _get_varint(handle, j)
        items += list(_block_entries(_read_block(d, o, s, True)))
    return items


HEADER = b"\x08\x01" + b"\x1a\x02\x08\x01"      # BundleHeaderProto: num_shards = 1, (endianness LITTLE = default, omitted), version { producer: 1 }


def write_bundle(prefix, tensors):
    """{name: float32 array} -> <prefix>.index + <prefix>.data-00000-of-00001 (names in byte order, as TensorFlow's BundleWriter does)."""
    names = sorted(tensors, key=lambda s: s.encode())
    entries, off = [], 0
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for n in names:
            a = np.asarray(tensors[n], "<f4")                  # (ascontiguousarray would turn the scalar beta powers into shape (1,))
            a = a if a.flags.c_contiguous else a.copy()
            raw = a.tobytes()
            f.write(raw)
            entries.append((n.encode(), Entry(DT_FLOAT, a.shape, 0, off, len(raw), mask(crc32c(raw))).serialize()))
            off += len(raw)
    with open(prefix + ".index", "wb") as f:
        f.write(build_table([(b"", HEADER)] + entries))


def read_bundle(prefix, verify=True):
    """-> {name: float32 array} (every DT_FLOAT tensor; each tensor's crc32c checked like TensorFlow's BundleReader does)."""
    _, entries = read_index(prefix + ".index", verify)
    data_path = prefix + ".data-00000-of-00001"
    if not os.path.exists(data_path):
        raise FileNotFoundError(f"{data_path}: the bundle's data file is missing (the index alone holds names and shapes only)")
    out = {}
    with open(data_path, "rb") as f:
        for name, e in entries.items():
            if e.dtype != DT_FLOAT:
                continue
            f.seek(e.offset)
            a = np.frombuffer(f.read(e.size), "<f4")
            if a.size * 4 != e.size or a.size != int(np.prod(e.shape, dtype=np.int64)):
                raise ValueError(f"{name}: {e.size} bytes do not make a float32{list(e.shape)}")
            if verify and mask(crc32c(a)) != e.crc32c:
                raise ValueError(f"{name}: checksum mismatch in {data_path}")
            out[name] = a.reshape(e.shape).copy()
    return out


# ------------------------------------------------------------------------------------------------ variable names <-> flat vectors
def _var(i):
    return "Variable" if i == 0 else f"Variable_{i}"


def layout(two_nets):
    """(online names, target names or None, Adam m names, Adam v names, beta power names) for the reference's two families of graphs."""
    if not two_nets:
        on = [_var(i) for i in range(10)]
        return on, None, [n + "/Adam" for n in on], [n + "/Adam_1" for n in on], ("beta1_power", "beta2_power")
    on = ["eval_net/" + _var(i) for i in range(10)]
    tg = ["target_net/" + _var(i) for i in range(10)]
    return (on, tg, ["target_net/" + n + "/Adam" for n in on], ["target_net/" + n + "/Adam_1" for n in on],
            ("target_net/beta1_power", "target_net/beta2_power"))


def _flat(tensors, names):
    parts = []
    for n, shp in zip(names, SHAPES):
        a = tensors[n]
        if tuple(a.shape) != shp:
            raise ValueError(f"{n}: shape {a.shape}, expected {shp}")
        parts.append(np.asarray(a, np.float32).ravel())
    return np.concatenate(parts)


def load_flat(prefix, verify=True):
    """A reference checkpoint as this framework's flat vectors:
    -> dict(online, target (or None), adam_m, adam_v, beta_pows (float32[2]) -- the last three None when the optimizer was not saved)."""
    t = read_bundle(prefix, verify)
    two = "eval_net/Variable" in t
    on, tg, m, v, pows = layout(two)
    out = {"online": _flat(t, on), "target": _flat(t, tg) if tg else None, "adam_m": None, "adam_v": None, "beta_pows": None}
    if all(n in t for n in m + v + list(pows)):
        out["adam_m"], out["adam_v"] = _flat(t, m), _flat(t, v)
        out["beta_pows"] = np.array([t[pows[0]].reshape(()), t[pows[1]].reshape(())], np.float32)
    return out


def save_flat(prefix, online, target=None, adam_m=None, adam_v=None, beta_pows=None):
    """The inverse: flat vectors -> a bundle with the reference's variable names (two-net names when `target` is given)."""
    on, tg, m, v, pows = layout(target is not None)

    def split(flat, names):
        flat = np.asarray(flat, np.float32).ravel()
        if flat.size != sum(int(np.prod(s)) for s in SHAPES):
            raise ValueError(f"expected {sum(int(np.prod(s)) for s in SHAPES)} parameters (plain head, 512 units, 2 actions), got {flat.size}")
        out, o = {}, 0
        for n, shp in zip(names, SHAPES):
            k = int(np.prod(shp))
            out[n] = flat[o:o + k].reshape(shp)
            o += k
        return out

    tensors = split(online, on)
    if tg:
        tensors.update(split(target, tg))
    if adam_m is not None:
        tensors.update(split(adam_m, m))
        tensors.update(split(adam_v, v))
        tensors[pows[0]] = np.float32(beta_pows[0]).reshape(())
        tensors[pows[1]] = np.float32(beta_pows[1]).reshape(())
    write_bundle(prefix, tensors)
