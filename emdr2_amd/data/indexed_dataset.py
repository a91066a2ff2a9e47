"""Reader / writer of the memory-mapped token datasets the reference keeps its evidence in (--indexed-evidence-data-path,
--indexed-title-data-path; megatron/data/indexed_dataset.py:335-513, `MMapIndexedDataset`).

File pair `<prefix>.idx` / `<prefix>.bin`:
    .idx  magic b'MMIDIDX\\0\\0' | u64 version = 1 | u8 dtype code | u64 n_sequences | u64 n_documents |
          int32 sizes[n_sequences] | int64 byte pointers[n_sequences] | int64 doc_idx[n_documents]
    .bin  the token arrays back to back
dtype codes: 1 uint8, 2 int8, 3 int16, 4 int32, 5 int64, 6 float64 (`np.float` in the reference), 7 float64, 8 uint16.
Files written by the reference's `tools/create_evidence_indexed_dataset.py` load here unchanged and vice versa (tests/golden/mmap_ref.*).
"""
import struct

import numpy as np

_HDR_MAGIC = b'MMIDIDX\x00\x00'
DTYPES = {1: np.uint8, 2: np.int8, 3: np.int16, 4: np.int32, 5: np.int64, 6: np.float64, 7: np.float64, 8: np.uint16}


def _code(dtype):
    dtype = np.dtype(dtype)
    for k in (1, 2, 3, 4, 5, 7, 8):
        if np.dtype(DTYPES[k]) == dtype:
            return k
    raise ValueError(dtype)


def index_file_path(prefix_path):
    return prefix_path + '.idx'


def data_file_path(prefix_path):
    return prefix_path + '.bin'


class MMapIndexedDataset(object):
    def __init__(self, path, skip_warmup=True):
        with open(index_file_path(path), 'rb') as f:
            if f.read(9) != _HDR_MAGIC:
                raise ValueError("not an MMapIndexedDataset index: %s" % index_file_path(path))
            if struct.unpack('<Q', f.read(8)) != (1,):
                raise ValueError("unsupported index version")
            self._dtype = np.dtype(DTYPES[struct.unpack('<B', f.read(1))[0]])
            self._len, self._doc_count = struct.unpack('<Q', f.read(8))[0], struct.unpack('<Q', f.read(8))[0]
            offset = f.tell()
        self._idx = np.memmap(index_file_path(path), mode='r', order='C')
        self.sizes = np.frombuffer(self._idx, dtype=np.int32, count=self._len, offset=offset)
        self.pointers = np.frombuffer(self._idx, dtype=np.int64, count=self._len, offset=offset + self.sizes.nbytes)
        self.doc_idx = np.frombuffer(self._idx, dtype=np.int64, count=self._doc_count, offset=offset + self.sizes.nbytes + self.pointers.nbytes)
        self._bin = np.memmap(data_file_path(path), mode='r', order='C')

    dtype = property(lambda self: self._dtype)

    def __len__(self):
        return self._len

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += self._len
        return np.frombuffer(self._bin, dtype=self._dtype, count=int(self.sizes[i]), offset=int(self.pointers[i]))

    def get(self, idx, offset=0, length=None):
        """A slice of one sequence (indexed_dataset.py:500-511)."""
        n = int(self.sizes[idx]) - offset if length is None else length
        return np.frombuffer(self._bin, dtype=self._dtype, count=n, offset=int(self.pointers[idx]) + offset * self._dtype.itemsize)

    def flat_tokens(self):
        """All tokens as one array + int64 sequence offsets (element units): what `EvidenceArena.from_flat` uploads without a Python loop.
        Valid when the sequences are stored back to back in order, which is how every builder writes them."""
        off = np.zeros(self._len + 1, dtype=np.int64)
        np.cumsum(self.sizes, out=off[1:])
        if self._len and not np.array_equal(self.pointers, off[:-1] * self._dtype.itemsize):
            raise ValueError("sequences are not stored contiguously")
        return np.frombuffer(self._bin, dtype=self._dtype, count=int(off[-1])), off


class MMapIndexedDatasetBuilder(object):
    def __init__(self, out_file, dtype=np.uint16):
        self._data = open(out_file, 'wb')
        self._dtype = np.dtype(dtype)
        self._sizes, self._doc_idx = [], [0]

    def add_item(self, tokens):
        a = np.asarray(tokens, dtype=self._dtype)
        self._data.write(a.tobytes(order='C'))
        self._sizes.append(a.size)

    def end_document(self):
        self._doc_idx.append(len(self._sizes))

    def finalize(self, index_file):
        self._data.close()
        sizes = np.asarray(self._sizes, dtype=np.int32)
        pointers = np.zeros(len(sizes), dtype=np.int64)
        if len(sizes) > 1:
            np.cumsum(sizes[:-1].astype(np.int64) * self._dtype.itemsize, out=pointers[1:])
        with open(index_file, 'wb') as f:
            f.write(_HDR_MAGIC)
            f.write(struct.pack('<Q', 1))
            f.write(struct.pack('<B', _code(self._dtype)))
            f.write(struct.pack('<Q', len(sizes)))
            f.write(struct.pack('<Q', len(self._doc_idx)))
            f.write(sizes.tobytes(order='C'))
            f.write(pointers.tobytes(order='C'))
            f.write(np.asarray(self._doc_idx, dtype=np.int64).tobytes(order='C'))


def make_dataset(path, impl='mmap', skip_warmup=True):
    """`make_dataset` of the reference for the only implementation its scripts use (--data-impl mmap / infer)."""
    if impl not in ('mmap', 'infer'):
        raise NotImplementedError("only the mmap implementation is supported (the reference scripts use it)")
    return MMapIndexedDataset(path, skip_warmup)
