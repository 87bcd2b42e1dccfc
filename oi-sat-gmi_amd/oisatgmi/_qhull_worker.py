"""Worker process of ``interpolator.interpolator_many``: triangulates granules on the host, nothing else.

Started as ``python -m oisatgmi._qhull_worker`` (a child process with pipes -- not multiprocessing: "spawn" re-imports the
caller's main script, and the reference's run/job.py has no ``__main__`` guard; "fork" must not be used from a process
that holds a GPU context).  Protocol, both ways: 8-byte little-endian length, then a pickle.  Request: ``(lon, lat)``
float arrays of the pixel centres; reply: the ``scipy.spatial.Delaunay`` of them as the reference builds it
(interpolator.py:151-155) with ``transform`` / ``vertex_to_simplex`` / ``neighbors`` filled in, or ``None`` when qhull
fails.  Ends at end of input.  Never imports the HIP binding or torch."""
import pickle
import struct
import sys


def triangulate(lon, lat):
    import numpy as np
    from scipy.spatial import Delaunay
    pts = np.column_stack((np.ravel(lon), np.ravel(lat))).astype(np.float64)
    try:
        tri = Delaunay(pts)
        tri.transform, tri.vertex_to_simplex, tri.neighbors      # noqa: B018 -- lazily computed members, carried by the pickle
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


def main():
    fin, fout = sys.stdin.buffer, sys.stdout.buffer
    while True:
        req = read_msg(fin)
        if req is None:
            return
        write_msg(fout, triangulate(*req))


if __name__ == "__main__":
    main()
