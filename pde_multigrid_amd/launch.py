"""Rank start-up and control plane for N > 1 WITHOUT PyTorch (one node, one process per GPU).

The data plane of the slab decomposition is RCCL inside libmgx (csrc/mgx_comm.hip).  What the ranks need besides it is
tiny: the 128-byte ncclUniqueId from rank 0, a barrier, a maximum over ranks and a gather of small objects.  Importing
torch for that made the whole N > 1 run bind to the ROCm runtime and RCCL that torch bundles (another version than the
one libmgx was built with).  This module does it with one TCP connection per rank to rank 0 on 127.0.0.1:

  * started by a launcher that sets RANK / LOCAL_RANK / WORLD_SIZE (the driver uses `python -m torch.distributed.run`;
    only its environment is used, MASTER_PORT itself belongs to its own store): rank 0 listens on an ephemeral port and
    publishes it in a file named after the job (MASTER_ADDR, MASTER_PORT, TORCHELASTIC_RUN_ID), the others poll it;
  * started plainly (`bench.py --gpus N`): `spawn` counts the GPUs without initialising one, starts N fresh children
    with that environment plus a random job key, watches them, and kills the group if one fails or the job times out.

Every wait has a deadline; a rank that fails closes its sockets, which fails the others' next collective.
"""
import base64
import glob
import json
import os
import socket
import struct
import subprocess
import sys
import tempfile
import time
import uuid


class RendezvousError(RuntimeError):
    pass


def job_key(env=None):
    env = os.environ if env is None else env
    if env.get("MGX_RDZV_KEY"):
        return env["MGX_RDZV_KEY"]
    return "tr_%s_%s_%s" % (env.get("MASTER_ADDR", "local"), env.get("MASTER_PORT", "0"), env.get("TORCHELASTIC_RUN_ID", "none"))


def _send(sock, obj):
    data = json.dumps(obj).encode()
    sock.sendall(struct.pack("!I", len(data)) + data)


def _recv(sock):
    head = b""
    while len(head) < 4:
        chunk = sock.recv(4 - len(head))
        if not chunk:
            raise RendezvousError("peer closed the connection")
        head += chunk
    (n,) = struct.unpack("!I", head)
    data = b""
    while len(data) < n:
        chunk = sock.recv(min(1 << 16, n - len(data)))
        if not chunk:
            raise RendezvousError("peer closed the connection")
        data += chunk
    return json.loads(data.decode())


class Rendezvous:
    """all_gather / broadcast / barrier / max over `world` processes of one node, star topology through rank 0"""

    def __init__(self, rank, world, key=None, timeout=120.0):
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        self.key = key or job_key()
        self.path = os.path.join(tempfile.gettempdir(), "mgx_rdzv_%s" % "".join(c if c.isalnum() or c in "._-" else "_" for c in self.key))
        self.peers, self.sock, self.server = {}, None, None
        if self.world <= 1:
            return
        deadline = time.monotonic() + self.timeout
        if self.rank == 0:
            self.server = socket.socket()
            self.server.bind(("127.0.0.1", 0))
            self.server.listen(self.world)
            self.server.settimeout(1.0)
            tmp = "%s.%d" % (self.path, os.getpid())
            with open(tmp, "w") as fh:
                fh.write("%d\n" % self.server.getsockname()[1])
            os.replace(tmp, self.path)  # atomic: a reader sees the old job's file or this one, never half of one
            while len(self.peers) < self.world - 1:
                if time.monotonic() > deadline:
                    raise RendezvousError("rendezvous: %d of %d ranks connected in %.0f s" % (len(self.peers) + 1, self.world, self.timeout))
                try:
                    c, _ = self.server.accept()
                except socket.timeout:
                    continue
                c.settimeout(self.timeout)
                try:
                    hello = _recv(c)
                except (RendezvousError, OSError, ValueError):
                    c.close()
                    continue
                if hello.get("key") != self.key or not 0 < int(hello.get("rank", -1)) < self.world or int(hello["rank"]) in self.peers:
                    c.close()
                    continue
                _send(c, {"key": self.key})
                self.peers[int(hello["rank"])] = c
        else:
            while True:
                if time.monotonic() > deadline:
                    raise RendezvousError("rendezvous: rank %d found no rank 0 for job %r in %.0f s" % (self.rank, self.key, self.timeout))
                try:
                    with open(self.path) as fh:
                        port = int(fh.read().strip())
                    s = socket.create_connection(("127.0.0.1", port), timeout=2.0)
                    s.settimeout(self.timeout)
                    _send(s, {"key": self.key, "rank": self.rank})
                    if _recv(s).get("key") == self.key:
                        self.sock = s
                        break
                    s.close()
                except (OSError, ValueError, RendezvousError):
                    pass  # no file yet, a stale file of an earlier job, or somebody else's port
                time.sleep(0.05)

    def all_gather(self, obj):
        """list of every rank's JSON-serialisable object, in rank order, on every rank"""
        if self.world <= 1:
            return [obj]
        try:
            if self.rank == 0:
                parts = [obj] + [None] * (self.world - 1)
                for r, c in self.peers.items():
                    parts[r] = _recv(c)
                for c in self.peers.values():
                    _send(c, parts)
                return parts
            _send(self.sock, obj)
            return _recv(self.sock)
        except (OSError, ValueError) as e:
            raise RendezvousError("collective failed on rank %d: %r" % (self.rank, e))

    def barrier(self):
        self.all_gather(None)

    def broadcast(self, obj, src=0):
        return self.all_gather(obj if self.rank == src else None)[src]

    def broadcast_bytes(self, data, src=0):
        got = self.broadcast(base64.b64encode(data).decode() if self.rank == src else None, src)
        return base64.b64decode(got)

    def max(self, x):
        return max(self.all_gather(float(x)))

    def close(self):
        for c in self.peers.values():
            c.close()
        if self.sock:
            self.sock.close()
        if self.server:
            self.server.close()
            try:
                os.unlink(self.path)
            except OSError:
                pass
        self.peers, self.sock, self.server = {}, None, None


def count_gpus():
    """GPUs of this node WITHOUT initialising the HIP runtime in this process: the KFD topology in sysfs (an agent with
    SIMDs is a GPU), restricted by ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES; if sysfs is not there, a throw-away child
    asks the runtime"""
    n = 0
    for props in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            with open(props) as fh:
                for line in fh:
                    k, _, v = line.partition(" ")
                    if k == "simd_count" and int(v) > 0:
                        n += 1
        except (OSError, ValueError):
            pass
    if n == 0:
        try:
            root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            out = subprocess.check_output([sys.executable, "-c", "import ctypes, sys; sys.path.insert(0, %r); import pde_multigrid_amd as P; "
                                           "c = ctypes.c_int(0); P.lib.mgx_device_count(ctypes.byref(c)); print(c.value)" % root],
                                          timeout=120, stderr=subprocess.DEVNULL)
            n = int(out.decode().strip().splitlines()[-1])
        except (subprocess.SubprocessError, OSError, ValueError, IndexError):
            n = 0
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):
        if os.environ.get(var, "").strip():
            n = min(n, len([x for x in os.environ[var].split(",") if x.strip()]))
    return n


def spawn(script, argv, nranks, timeout=1500.0, env_extra=None, need_gpus=True):
    """start `nranks` fresh children `python script argv...` (RANK / LOCAL_RANK / WORLD_SIZE / MGX_RDZV_KEY in their
    environment), wait for them; the first failure or the deadline kills the others.  Returns the exit code.  This
    process never touches the GPU and never exec's."""
    if need_gpus:
        found = count_gpus()
        if found < nranks:
            sys.stderr.write("%s: --gpus %d needs %d GPUs, found %d\n" % (os.path.basename(script), nranks, nranks, found))
            return 2
    key = "job_%s" % uuid.uuid4().hex
    procs = []
    for r in range(nranks):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(nranks), "MGX_RDZV_KEY": key})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.update(env_extra or {})
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env))
    deadline = time.monotonic() + timeout
    rc = 0
    try:
        while procs and rc == 0:
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0:
                    rc = code if code > 0 else 1
            if time.monotonic() > deadline:
                sys.stderr.write("%s: the ranks did not finish within %.0f s\n" % (os.path.basename(script), timeout))
                rc = 124
            time.sleep(0.05)
    finally:
        for p in procs:  # exactly the children started here
            p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    return rc
