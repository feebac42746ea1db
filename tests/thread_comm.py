"""A thread communicator for include/mdbg_dist.h written in Python (ctypes callbacks): W ranks = W threads of this process sharing one GPU, the
exchange is device-to-device copies.  Test infrastructure (tests/test_gpu_dist_scale.py); scratch/measure_dist_traffic.py carries its own copy."""
import ctypes as C
import threading

import numpy as np

from rust_mdbg_amd import dist_c


class Xfer(C.Structure):            # mdbg_xfer
    _fields_ = [("peer", C.c_uint32), ("d_ptr", C.c_void_p), ("bytes", C.c_uint64)]


AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_uint64))
EX = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(Xfer), C.c_uint32, C.POINTER(Xfer), C.c_uint32)
AR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64)
D2D, D2H, H2D = 3, 2, 1


class ThreadWorld:
    def __init__(self, world):
        self.W = world
        self.bar = threading.Barrier(world)
        self.ag = [None] * world
        self.posted = [None] * world
        self.red = [None] * world
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

    def comm(self, rank):
        """-> (mdbg_comm for `rank`, the callback objects: keep them alive as long as the mdbg_dist)"""
        w, W, hip = self, self.W, self.hip

        def allgather(_, send, n, recv):
            w.ag[rank] = [send[i] for i in range(n)]
            w.bar.wait()
            for r in range(W):
                for i in range(n):
                    recv[r * n + i] = w.ag[r][i]
            w.bar.wait()
            return 0

        def exchange(_, sends, ns, recvs, nr):
            w.posted[rank] = [(sends[i].peer, sends[i].d_ptr, sends[i].bytes) for i in range(ns)]
            w.bar.wait()
            nxt = {}
            for i in range(nr):
                p = recvs[i].peer
                mine = [x for x in w.posted[p] if x[0] == rank]          # transfers between a pair are matched in the order they are listed
                j = nxt.get(p, 0)
                nxt[p] = j + 1
                if j >= len(mine) or mine[j][2] != recvs[i].bytes:
                    return -3
                if hip.hipMemcpy(recvs[i].d_ptr, mine[j][1], recvs[i].bytes, D2D) != 0:
                    return -4
            if hip.hipDeviceSynchronize() != 0:
                return -4
            w.bar.wait()
            return 0

        def allreduce(_, d_buf, n):
            a = np.empty(n, dtype=np.uint64)
            if n and hip.hipMemcpy(a.ctypes.data, d_buf, n * 8, D2H) != 0:
                return -4
            w.red[rank] = a
            w.bar.wait()
            s = w.red[0].copy()
            for r in range(1, W):
                s += w.red[r]
            w.bar.wait()
            if n and hip.hipMemcpy(d_buf, s.ctypes.data, n * 8, H2D) != 0:
                return -4
            return 0

        fns = (AG(allgather), EX(exchange), AR(allreduce))
        cm = dist_c.Comm()
        cm.self = None
        cm.rank = rank
        cm.world = W
        cm.allgather_u64 = C.cast(fns[0], C.c_void_p)
        cm.exchange = C.cast(fns[1], C.c_void_p)
        cm.allreduce_sum_u64 = C.cast(fns[2], C.c_void_p)
        return cm, fns
