"""Worker process of ``interpolator.interpolator_many``: triangulates granules on the host, nothing else.

Started as ``python -m oisatgmi._qhull_worker`` (a child process with pipes -- not multiprocessing: "spawn" re-imports the
caller's main script, and the reference's run/job.py has no ``__main__`` guard; "fork" must not be used from a process
that holds a GPU context).  Protocol, both ways: 8-byte little-endian length, then a pickle.  Request: ``(lon, lat)``
float arrays of the pixel centres; reply: ``("ok", tri)`` with the ``scipy.spatial.Delaunay`` of them as the reference builds
it (interpolator.py:151-155), ``vertex_to_simplex`` / ``neighbors`` filled in, or ``("failed", None)`` when
qhull refuses the points (the reference then skips the granule).  A reply that never comes -- the pipe at end of file --
means the worker died, which is an error and not a skipped granule.  Ends at end of input.  Never loads the HIP library or
torch; whatever a library prints goes to stderr, the reply pipe carries replies only."""
import os
import pickle
import struct
import sys


class WorkerDied(RuntimeError):
    pass


def triangulate(lon, lat):
    import numpy as np
    from scipy.spatial import Delaunay
    pts = np.column_stack((np.ravel(lon), np.ravel(lat))).astype(np.float64)
    try:
        tri = Delaunay(pts)
        tri.vertex_to_simplex, tri.neighbors      # noqa: B018 -- lazily computed members, carried by the pickle (the barycentric
        #                                           transforms are the device's job: oisat_tri_transform)
        return tri
    except Exception:
        return None


def read_msg(f):
    head = f.read(8)
    if len(head) < 8:
        return None
    (n,) = struct.unpack("<Q", head)
    return pickle.loads(f.read(n))


def write_msg(f, obj):
    blob = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    f.write(struct.pack("<Q", len(blob)))
    f.write(blob)
    f.flush()


def reply_of(f):
    """The parent's side of one exchange: the triangulation, ``None`` if qhull failed, ``WorkerDied`` at end of file."""
    msg = read_msg(f)
    if msg is None:
        raise WorkerDied("a triangulation worker ended without replying")
    status, tri = msg
    return tri if status == "ok" else None


def main():
    fin = sys.stdin.buffer
    fout = os.fdopen(os.dup(sys.stdout.fileno()), "wb")     # the reply pipe, out of reach of print()
    os.dup2(sys.stderr.fileno(), sys.stdout.fileno())
    sys.stdout = sys.stderr
    while True:
        req = read_msg(fin)
        if req is None:
            return
        tri = triangulate(*req)
        write_msg(fout, ("ok", tri) if tri is not None else ("failed", None))


if __name__ == "__main__":
    main()
