"""Reader/writer of the harness' PointCloud2 file format (host/pointcloud2.cpp) for tests and tools."""
import numpy as np

FLOAT32 = 7


def write_pointcloud2(path, xyz, point_step=16, offsets=(0, 4, 8), frame_id="kinect2_link", is_dense=True,
                      width=None, height=1, extra_fields=()):
    """xyz (n,3) float32 -> file with FLOAT32 fields x,y,z at `offsets` inside `point_step`-byte points."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    n = len(xyz)
    width = n // height if width is None else width
    assert width * height == n
    data = np.zeros((n, point_step), dtype=np.uint8)
    for a, off in enumerate(offsets):
        data[:, off:off + 4] = xyz[:, a].copy().view(np.uint8).reshape(n, 4)
    fields = [("x", offsets[0], FLOAT32, 1), ("y", offsets[1], FLOAT32, 1), ("z", offsets[2], FLOAT32, 1)] + list(extra_fields)
    with open(path, "wb") as f:
        hdr = "PCOP_POINTCLOUD2 1\nframe_id %s\nheight %d\nwidth %d\nis_bigendian 0\npoint_step %d\nrow_step %d\nis_dense %d\nfields %d\n" % (
            frame_id or "-", height, width, point_step, point_step * width, 1 if is_dense else 0, len(fields))
        for nm, off, dt, cnt in fields:
            hdr += "%s %d %d %d\n" % (nm, off, dt, cnt)
        hdr += "data %d\n" % data.size
        f.write(hdr.encode())
        f.write(data.tobytes())


def read_pointcloud2(path):
    """-> dict(xyz (n,3) float32, frame_id, width, height, point_step, is_dense, fields)."""
    with open(path, "rb") as f:
        raw = f.read()
    pos = 0

    def line():
        nonlocal pos
        e = raw.index(b"\n", pos)
        s = raw[pos:e].decode()
        pos = e + 1
        return s

    assert line().startswith("PCOP_POINTCLOUD2")
    kv = {}
    for key in ("frame_id", "height", "width", "is_bigendian", "point_step", "row_step", "is_dense", "fields"):
        k, v = line().split(" ", 1)
        assert k == key
        kv[k] = v
    fields = []
    for _ in range(int(kv["fields"])):
        nm, off, dt, cnt = line().split()
        fields.append((nm, int(off), int(dt), int(cnt)))
    nbytes = int(line().split()[1])
    data = np.frombuffer(raw, dtype=np.uint8, count=nbytes, offset=pos)
    w, h, step, row = int(kv["width"]), int(kv["height"]), int(kv["point_step"]), int(kv["row_step"])
    pts = data.reshape(h, row)[:, :w * step].reshape(h * w, step)
    off = {nm: o for nm, o, dt, _ in fields if dt == FLOAT32}
    xyz = np.stack([pts[:, off[a]:off[a] + 4].copy().view(np.float32).reshape(-1) for a in "xyz"], axis=1)
    return dict(xyz=xyz, frame_id=kv["frame_id"], width=w, height=h, point_step=step, is_dense=bool(int(kv["is_dense"])),
                fields=fields)
