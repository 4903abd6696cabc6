"""The load path of the C ABI (csrc/stloader.hip): exl2_stloader_read, exl2_tensor_remap, exl2_tensor_remap_4bit
against the reference's semantics (ext_stloader.cpp:11-219: the bytes of [offset, offset + size) end up in the tensor; the
column re-orderings are the loops of :176-182 and :203-216, restated with numpy indexing here).

Host targets run on the emulation backend AND on the real library without a GPU (the host branch makes no HIP call); the
device branch (pinned ring, chunked copies, slot reuse past 16 chunks) runs on the emulation backend, where the "device" is
host memory, and on the MI355X (-m gpu)."""
import os

import numpy as np
import pytest
import torch

MIB = 1 << 20


@pytest.fixture(scope="module")
def blob(tmp_path_factory):
    """70 MiB + 123 bytes: more than the 16 x 4 MiB pinned ring, ragged end"""
    p = tmp_path_factory.mktemp("st") / "blob.bin"
    rng = np.random.default_rng(7)
    data = rng.integers(0, 256, size=70 * MIB + 123, dtype=np.uint8)
    data.tofile(p)
    return str(p), data


def _real_lib_ext():
    """the product library bound without a GPU: only its host-only entry points may be called"""
    from exllamav2_amd.ext import ExtC
    e = ExtC(allow_cpu=True)
    try:
        e.lib
    except Exception as ex:
        pytest.skip(f"libexl2_hip.so not loadable here: {ex}")
    return e


@pytest.mark.parametrize("offset,size", [(0, 1), (5, 4 * MIB), (4096, 4 * MIB + 1), (17, 33 * MIB + 5), (0, 70 * MIB + 123)])
def test_read_into_host_tensor(be, blob, offset, size):
    path, data = blob
    t = torch.zeros((size,), dtype=torch.uint8)
    be.ext.stloader_read(path, offset, size, t)
    assert np.array_equal(t.numpy(), data[offset:offset + size])


def test_read_into_host_tensor_real_library_without_gpu(blob):
    e = _real_lib_ext()
    path, data = blob
    t = torch.zeros((9 * MIB + 2) // 2, dtype=torch.float16)
    e.stloader_read(path, 3, 9 * MIB + 2, t)
    assert np.array_equal(t.numpy().view(np.uint8), data[3:3 + 9 * MIB + 2])
    # column re-orderings through the real library as well (host only)
    _check_remaps(e)


@pytest.mark.parametrize("offset,size", [(1, 7), (0, 4 * MIB), (9, 13 * MIB + 3), (64, 70 * MIB)])
def test_read_into_device_tensor(be, blob, offset, size):
    """device branch: chunked through the pinned ring (70 MiB = 18 chunks > 16 slots: slots are handed back and reused)"""
    path, data = blob
    t = torch.zeros((size,), dtype=torch.uint8, device=be.device)
    if be.is_emu:
        # the emulation library treats target_device >= 0 like a device (memcpy is the emulated copy)
        be.ext.lib.check(be.ext.lib.exl2_stloader_read(os.fsencode(path), offset, size, t.data_ptr(), 0, None))
    else:
        be.ext.stloader_read(path, offset, size, t)
    assert np.array_equal(be.n(t), data[offset:offset + size])
    # a second call reuses the ring
    t2 = torch.zeros((5 * MIB,), dtype=torch.uint8, device=be.device)
    if be.is_emu:
        be.ext.lib.check(be.ext.lib.exl2_stloader_read(os.fsencode(path), 11, 5 * MIB, t2.data_ptr(), 0, None))
    else:
        be.ext.stloader_read(path, 11, 5 * MIB, t2)
    assert np.array_equal(be.n(t2), data[11:11 + 5 * MIB])


def test_read_errors(be, blob, tmp_path):
    path, data = blob
    t = torch.zeros((64,), dtype=torch.uint8)
    with pytest.raises(RuntimeError, match="cannot open"):
        be.ext.stloader_read(str(tmp_path / "missing.bin"), 0, 64, t)
    with pytest.raises(RuntimeError, match="I/O error"):
        be.ext.stloader_read(path, data.size - 10, 64, t)              # range runs past the end of the file
    with pytest.raises(RuntimeError, match="bytes requested"):
        be.ext.stloader_read(path, 0, 60, t)                           # size != tensor bytes
    with pytest.raises(RuntimeError, match="contiguous"):
        be.ext.stloader_read(path, 0, 32, torch.zeros((8, 8), dtype=torch.uint8)[:, :4])
    be.ext.stloader_read(path, 0, 0, torch.zeros((0,), dtype=torch.uint8))      # empty tensor: no-op (stloader.py:160 calls it)
    if be.is_emu:                                                      # device branch, short file
        with pytest.raises(RuntimeError, match="I/O error"):
            be.ext.lib.check(be.ext.lib.exl2_stloader_read(os.fsencode(path), data.size - 10, 64, t.data_ptr(), 0, None))


def _check_remaps(ext):
    rng = np.random.default_rng(0)
    for rows, cols in ((1, 8), (5, 64), (3, 4096), (0, 16)):
        t = rng.integers(-2 ** 31, 2 ** 31 - 1, size=(rows, cols), dtype=np.int64).astype(np.int32)
        idx = rng.permutation(cols).astype(np.int32)
        tt = torch.from_numpy(t.copy())
        ext.tensor_remap(tt, torch.from_numpy(idx))
        assert np.array_equal(tt.numpy(), t[:, idx])                   # ext_stloader.cpp:176-182: *a++ = temp[idx[c]]
        q = rng.integers(-2 ** 31, 2 ** 31 - 1, size=(rows, cols // 8), dtype=np.int64).astype(np.int32)
        u = q.view(np.uint32)
        nib = np.stack([(u >> (4 * b)) & 0xF for b in range(8)], axis=-1).reshape(rows, cols)
        want_n = nib[:, idx].reshape(rows, cols // 8, 8)               # ext_stloader.cpp:203-216: nibble c <- nibble idx[c]
        want = np.zeros((rows, cols // 8), dtype=np.uint32)
        for b in range(8):
            want |= want_n[:, :, b].astype(np.uint32) << np.uint32(4 * b)
        qt = torch.from_numpy(q.copy())
        ext.tensor_remap_4bit(qt, torch.from_numpy(idx))
        assert np.array_equal(qt.numpy().view(np.uint32), want)
    # a non-permutation index is legal (the reference gathers), an out-of-range one is refused instead of read
    t = torch.arange(8, dtype=torch.int32).view(1, 8)
    ext.tensor_remap(t, torch.tensor([0, 0, 1, 1, 2, 2, 3, 3], dtype=torch.int32))
    assert t.tolist() == [[0, 0, 1, 1, 2, 2, 3, 3]]
    with pytest.raises(RuntimeError, match="outside"):
        ext.tensor_remap(t, torch.tensor([0, 1, 2, 3, 4, 5, 6, 8], dtype=torch.int32))
    with pytest.raises(RuntimeError):
        ext.tensor_remap(t, torch.zeros((7,), dtype=torch.int32))      # shape contract (TORCH_CHECK_SHAPES)
    with pytest.raises(RuntimeError):
        ext.tensor_remap(t.to(torch.int64), torch.zeros((8,), dtype=torch.int32))


def test_tensor_remaps():
    from tests.conftest import Backend
    _check_remaps(Backend("emu").ext)
