"""Device context and device arrays over the C-ABI (no PyTorch)."""
import ctypes as C
import os

import numpy as np

from dodt_amd import _lib


# ROCm maps a process's HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The frame-pair
# pipeline runs four streams (two conv stacks, two frames' prep + tail) and the multi-GPU exchange step
# adds a fifth: a stream that shares a queue runs behind the other's launches (measured with a 1-rank
# communicator: 247 pairs/s on four queues, 280 on eight, 284 without the fifth stream).  Read by the HIP
# runtime when it initialises, i.e. at the first Context of the process.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')


class DeviceArray(object):
    """A typed, shaped view of device memory.  Owns the allocation unless it
    wraps an external pointer (e.g. a torch tensor's data_ptr())."""

    def __init__(self, ctx, shape, dtype, ptr=None):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self._owns = ptr is None
        if ptr is None:
            p = C.c_void_p()
            _lib.check(ctx.lib.dodt_malloc(ctx.handle, self.nbytes, C.byref(p)),
                       'dodt_malloc')
            ptr = p.value
        self.ptr = ptr

    def offset(self, nbytes, shape, dtype=None):
        """View starting `nbytes` into this array (no ownership)."""
        return DeviceArray(self.ctx, shape, dtype or self.dtype,
                           ptr=self.ptr + int(nbytes))

    def upload(self, host):
        a = np.ascontiguousarray(host, dtype=self.dtype)
        if a.nbytes != self.nbytes:
            raise ValueError('upload: %d bytes into a %d-byte device array'
                             % (a.nbytes, self.nbytes))
        _lib.check(self.ctx.lib.dodt_memcpy_h2d(
            self.ctx.handle, self.ptr, a.ctypes.data, a.nbytes), 'h2d')
        return self

    def upload_async(self, pinned, ctx=None, nbytes=None):
        """Enqueue a copy from a PinnedArray on ctx's stream; the host does not wait."""
        c = ctx or self.ctx
        n = pinned.nbytes if nbytes is None else int(nbytes)
        if n > self.nbytes or n > pinned.nbytes:
            raise ValueError('upload_async: %d bytes do not fit' % n)
        _lib.check(c.lib.dodt_memcpy_h2d_async(c.handle, self.ptr, pinned.ptr, n), 'h2d async')
        return self

    def download(self):
        out = np.empty(self.shape, dtype=self.dtype)
        _lib.check(self.ctx.lib.dodt_memcpy_d2h(
            self.ctx.handle, out.ctypes.data, self.ptr, self.nbytes), 'd2h')
        return out

    def zero(self):
        _lib.check(self.ctx.lib.dodt_memset(self.ctx.handle, self.ptr, 0,
                                            self.nbytes), 'memset')
        return self

    def free(self):
        if self._owns and self.ptr:
            self.ctx.lib.dodt_free(self.ctx.handle, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedArray(object):
    """Page-locked host memory viewed as a numpy array (`.a`); the source of copies that do
    not block the host (DeviceArray.upload_async)."""

    def __init__(self, ctx, shape, dtype):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = C.c_void_p()
        _lib.check(ctx.lib.dodt_pinned_alloc(ctx.handle, self.nbytes, C.byref(p)),
                   'dodt_pinned_alloc')
        self.ptr = p.value
        buf = (C.c_char * max(self.nbytes, 1)).from_address(self.ptr)
        self.a = np.frombuffer(buf, dtype=self.dtype, count=int(np.prod(self.shape))).reshape(
            self.shape)

    def free(self):
        if self.ptr:
            self.a = None
            self.ctx.lib.dodt_pinned_free(self.ctx.handle, self.ptr)
            self.ptr = None


class Context(object):
    """One per GPU per process (include/dodt_hip.h conventions)."""

    def __init__(self, device_id=0, stream=None, high_priority=False):
        self.lib = _lib.load()
        h = C.c_void_p()
        if stream is None and high_priority:
            rc = self.lib.dodt_ctx_create_high_priority(int(device_id), C.byref(h))
        elif stream is None:
            rc = self.lib.dodt_ctx_create(int(device_id), C.byref(h))
        else:
            rc = self.lib.dodt_ctx_create_on_stream(int(device_id),
                                                    C.c_void_p(int(stream)),
                                                    C.byref(h))
        _lib.check(rc, 'dodt_ctx_create')
        self.handle = h
        self.device_id = int(device_id)

    def empty(self, shape, dtype=np.float32):
        return DeviceArray(self, shape, dtype)

    def zeros(self, shape, dtype=np.float32):
        return DeviceArray(self, shape, dtype).zero()

    def array(self, host, dtype=None):
        host = np.asarray(host)
        return DeviceArray(self, host.shape, dtype or host.dtype).upload(host)

    def pinned(self, shape, dtype=np.float32):
        return PinnedArray(self, shape, dtype)

    def wrap(self, ptr, shape, dtype=np.float32):
        return DeviceArray(self, shape, dtype, ptr=int(ptr))

    def sync(self):
        _lib.check(self.lib.dodt_ctx_sync(self.handle), 'dodt_ctx_sync')

    def wait_for(self, other):
        """Later work on this context waits for what `other` has enqueued so far."""
        _lib.check(self.lib.dodt_ctx_wait_for(self.handle, other.handle), 'dodt_ctx_wait_for')

    def mark(self, slot):
        _lib.check(self.lib.dodt_mark(self.handle, int(slot)), 'dodt_mark')

    def wait_mark(self, other, slot):
        """Later work on this context waits for `other`'s mark `slot` as last recorded."""
        _lib.check(self.lib.dodt_ctx_wait_mark(self.handle, other.handle, int(slot)), 'dodt_ctx_wait_mark')

    def elapsed_ms(self, slot, to_ctx, to_slot):
        """GPU time from this context's mark `slot` to `to_ctx`'s mark `to_slot`."""
        ms = C.c_float()
        _lib.check(self.lib.dodt_mark_elapsed(self.handle, int(slot), to_ctx.handle, int(to_slot),
                                              C.byref(ms)), 'dodt_mark_elapsed')
        return ms.value

    def timer_start(self):
        _lib.check(self.lib.dodt_timer_start(self.handle), 'timer_start')

    def timer_stop(self):
        ms = C.c_float()
        _lib.check(self.lib.dodt_timer_stop(self.handle, C.byref(ms)), 'timer_stop')
        return ms.value

    def close(self):
        if self.handle:
            self.lib.dodt_ctx_destroy(self.handle)
            self.handle = None


_default = {}


def default_context(device_id=0):
    if device_id not in _default:
        _default[device_id] = Context(device_id)
    return _default[device_id]
