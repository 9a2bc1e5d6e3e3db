"""File formats on either side of the reconstruction: particle readers and mesh writers with the
layouts of the reference's `splashsurf_lib::io` (citations relative to /root/reference/splashsurf_lib/src/io/).

    particles_from_file / particles_to_file     io.rs:17-43, splashsurf/src/io.rs:200-230
    mesh_from_file / mesh_to_file               splashsurf/src/io.rs:245-305

Formats: legacy VTK (`vtk_format.rs`, written like the vtkio crate does: version 4.2, BINARY big-endian,
UNSTRUCTURED_GRID), XML VTK `.vtu` (read only), raw little-endian XYZ triples (`xyz_format.rs`), PLY (`ply_format.rs`), OBJ (`obj_format.rs`),
BGEO v5 particles and point attributes, optionally gzip-compressed (`bgeo_format.rs`), JSON arrays (`json_format.rs`).
The writers reproduce the reference's files byte for byte (tests/golden/io/ holds files written by the reference).
Plain host-side Python/numpy: file IO is not on the GPU path.
"""
import gzip
import json
import os
import struct

import numpy as np


# ------------------------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------------------------
def _ext(path):
    e = os.path.splitext(str(path))[1].lower().lstrip(".")
    if not e:
        raise ValueError("unable to detect the file format (file name has to end with a supported extension)")
    return e


def _fmt_display(x):
    """Rust's `{}` for floats: shortest digits that round-trip, never an exponent, no trailing `.0`."""
    x = np.asarray(x)[()]
    if np.isnan(x):
        return "NaN"
    if np.isinf(x):
        return "inf" if x > 0 else "-inf"
    return np.format_float_positional(x, unique=True, trim="-")


def _fmt_json(x):
    """serde_json (ryu) for an f64: shortest round-trip digits, exponent without '+' and leading zeros."""
    s = repr(float(x))
    if "e" in s:
        m, e = s.split("e")
        sign = "-" if e.startswith("-") else ""
        s = m + "e" + sign + e.lstrip("+-").lstrip("0")
    return s


# ------------------------------------------------------------------------------------------------------------
# legacy VTK
# ------------------------------------------------------------------------------------------------------------
_VTK_TYPES = {"float": ">f4", "double": ">f8", "int": ">i4", "unsigned_int": ">u4", "long": ">i8", "unsigned_long": ">u8", "vtktypeint64": ">i8",
              "vtktypeint32": ">i4", "short": ">i2", "unsigned_short": ">u2", "char": ">i1", "unsigned_char": ">u1"}


class _VtkReader:
    def __init__(self, raw):
        self.raw = raw
        self.pos = 0

    def line(self):
        """next non-empty line (stripped), or None at end of file"""
        while self.pos < len(self.raw):
            e = self.raw.find(b"\n", self.pos)
            if e < 0:
                e = len(self.raw)
            ln = self.raw[self.pos:e].strip()
            self.pos = e + 1
            if ln:
                return ln.decode("ascii", "replace")
        return None

    def values(self, count, vtk_type, binary):
        dt = np.dtype(_VTK_TYPES[vtk_type.lower()])
        if binary:
            out = np.frombuffer(self.raw, dtype=dt, count=count, offset=self.pos)
            self.pos += count * dt.itemsize
            return out.astype(dt.newbyteorder("="))
        vals = []
        while len(vals) < count:
            ln = self.line()
            if ln is None:
                raise ValueError("unexpected end of VTK file")
            vals.extend(ln.split())
        if len(vals) != count:
            raise ValueError("malformed ASCII VTK section")
        return np.array([float(v) for v in vals]).astype(dt.newbyteorder("="))


def _read_vtk(path, points_only=False):
    """Legacy VTK (ASCII or BINARY) with an UNSTRUCTURED_GRID or POLYDATA data set -> dict(points, cells, point_data).
    points_only: stop after the POINTS section (enough for particles; also accepts version 5 files, whose CELLS layout
    this reader does not parse)."""
    raw = open(path, "rb").read()
    if raw.lstrip().startswith(b"<"):  # XML VTK under a .vtk name
        return _read_vtu(path)
    r = _VtkReader(raw)
    header = r.line()
    if header is None or not header.lower().startswith("# vtk datafile"):
        raise ValueError("not a legacy VTK file")
    # title line may be empty in the file; the reader above skips empty lines, so look for the format keyword
    ln = r.line()
    if ln is not None and ln.upper() not in ("ASCII", "BINARY"):
        ln = r.line()
    if ln is None or ln.upper() not in ("ASCII", "BINARY"):
        raise ValueError("VTK file: expected ASCII or BINARY")
    binary = ln.upper() == "BINARY"
    out = dict(points=None, cells=None, point_data={})
    n_points = 0
    n_cells_total = 0  # tuples of the arrays of a CELL_DATA section
    n_section = 0
    section = None
    while True:
        ln = r.line()
        if ln is None:
            break
        tok = ln.split()
        key = tok[0].upper()
        if key == "DATASET":
            if tok[1].upper() not in ("UNSTRUCTURED_GRID", "POLYDATA"):
                raise ValueError("VTK file does not contain supported data set pieces")
        elif key == "POINTS":
            n_points = int(tok[1])
            out["points"] = r.values(3 * n_points, tok[2], binary).reshape(n_points, 3)
            if points_only:
                return out
        elif key in ("CELLS", "POLYGONS", "VERTICES"):
            n_cells, size = int(tok[1]), int(tok[2])
            n_cells_total += n_cells
            vals = r.values(size, "int", binary)
            if key != "VERTICES":
                out["cells"] = (n_cells, vals)
        elif key == "CELL_TYPES":
            r.values(int(tok[1]), "int", binary)
        elif key == "POINT_DATA":
            section = "point"
            n_section = int(tok[1]) if len(tok) > 1 else n_points
        elif key == "CELL_DATA":
            section = "cell"  # arrays of a cell section hold one tuple per cell; they are read and discarded
            n_section = int(tok[1]) if len(tok) > 1 else n_cells_total
        elif key == "SCALARS":
            name, ty = tok[1], tok[2]
            ncomp = int(tok[3]) if len(tok) > 3 else 1
            save = r.pos
            nxt = r.line()
            if nxt is None or not nxt.upper().startswith("LOOKUP_TABLE"):
                r.pos = save
            n_tuples = n_section if section else n_points
            vals = r.values(ncomp * n_tuples, ty, binary)
            if section == "point":
                out["point_data"][name] = vals.reshape(n_points, ncomp) if ncomp > 1 else vals
        elif key in ("VECTORS", "NORMALS"):
            vals = r.values(3 * (n_section if section else n_points), tok[2], binary)
            if section == "point":
                out["point_data"][tok[1]] = vals.reshape(n_points, 3)
        elif key in ("METADATA", "INFORMATION", "FIELD", "OFFSETS", "CONNECTIVITY"):
            raise NotImplementedError("VTK section %s (file version 5 layouts) is not supported by this reader" % key)
        # anything else (e.g. a title line that was not skipped) is ignored
    if out["points"] is None:
        raise ValueError("VTK file has no POINTS section")
    return out


_VTU_TYPES = {"Float32": "f4", "Float64": "f8", "Int8": "i1", "UInt8": "u1", "Int16": "i2", "UInt16": "u2", "Int32": "i4", "UInt32": "u4",
              "Int64": "i8", "UInt64": "u8"}


def _read_vtu(path):
    """XML VTK UnstructuredGrid (`.vtu`, what vtkio reads for the reference: vtk_format.rs:40-140): first Piece; data arrays
    in ascii, inline base64 ("binary") or appended (raw / base64) form, 32- or 64-bit block headers, optional
    vtkZLibDataCompressor.  Returns dict(points, point_data) like _read_vtk."""
    import base64
    import re
    import xml.etree.ElementTree as ET
    import zlib
    raw = open(path, "rb").read()
    appended, enc = None, None
    m = re.search(rb"<AppendedData[^>]*>", raw)
    xml_part = raw
    if m:
        enc = re.search(rb'encoding="(\w+)"', m.group(0))
        enc = enc.group(1).decode() if enc else "base64"
        start = raw.index(b"_", m.end()) + 1
        end = raw.rindex(b"</AppendedData>")
        appended = raw[start:end]
        xml_part = raw[:m.start()] + raw[end + len(b"</AppendedData>"):]
    root = ET.fromstring(xml_part)
    if root.tag != "VTKFile" or root.get("type") != "UnstructuredGrid":
        raise ValueError("VTK file does not contain supported data set pieces")
    bo = "<" if root.get("byte_order", "LittleEndian") == "LittleEndian" else ">"
    hdr = bo + ("u8" if root.get("header_type", "UInt32") == "UInt64" else "u4")
    hsize = np.dtype(hdr).itemsize
    compressed = root.get("compressor") is not None
    if compressed and root.get("compressor") != "vtkZLibDataCompressor":
        raise NotImplementedError("VTU compressor %s is not supported" % root.get("compressor"))
    piece = root.find("./UnstructuredGrid/Piece")
    if piece is None:
        raise ValueError('VTK file does not contain a supported "piece".')
    n_points = int(piece.get("NumberOfPoints"))

    def decode_blocks(buf, is_base64):
        """One data array from `buf` (positioned at its header): returns the decoded payload bytes."""
        def take(nbytes, pos):
            if not is_base64:
                return buf[pos:pos + nbytes], pos + nbytes
            # base64 encodes header and payload as separate, individually padded streams
            nchar = ((nbytes + 2) // 3) * 4
            return base64.b64decode(buf[pos:pos + nchar])[:nbytes], pos + nchar
        if not compressed:
            h, pos = take(hsize, 0)
            n = int(np.frombuffer(h, dtype=hdr)[0])
            if is_base64:  # header and data share one base64 stream here
                nchar = ((hsize + n + 2) // 3) * 4
                return base64.b64decode(buf[:nchar])[hsize:hsize + n]
            return buf[pos:pos + n]
        h, pos = take(3 * hsize, 0)
        nblocks, _bsize, _last = (int(x) for x in np.frombuffer(h, dtype=hdr))
        if is_base64:  # the whole header (3 + nblocks words) is one base64 stream, the compressed blocks another
            hbytes = (3 + nblocks) * hsize
            nchar = ((hbytes + 2) // 3) * 4
            sizes = np.frombuffer(base64.b64decode(buf[:nchar])[3 * hsize:hbytes], dtype=hdr)
            data = base64.b64decode(buf[nchar:nchar + ((int(sizes.sum()) + 2) // 3) * 4])
            pos = 0
        else:
            sizes = np.frombuffer(buf[pos:pos + nblocks * hsize], dtype=hdr)
            data = buf
            pos += nblocks * hsize
        out = []
        for sz in sizes:
            out.append(zlib.decompress(data[pos:pos + int(sz)]))
            pos += int(sz)
        return b"".join(out)

    def array_of(da):
        ty = _VTU_TYPES.get(da.get("type"))
        if ty is None:
            raise ValueError("unsupported VTU data type %s" % da.get("type"))
        ncomp = int(da.get("NumberOfComponents", "1"))
        fmt = da.get("format", "ascii")
        if fmt == "ascii":
            vals = np.array((da.text or "").split(), dtype=np.float64).astype(np.dtype(ty))
        elif fmt == "binary":
            vals = np.frombuffer(decode_blocks("".join((da.text or "").split()).encode(), True), dtype=bo + ty)
        elif fmt == "appended":
            if appended is None:
                raise ValueError("VTU file refers to appended data but has no AppendedData section")
            off = int(da.get("offset", "0"))
            vals = np.frombuffer(decode_blocks(appended[off:], enc == "base64"), dtype=bo + ty)
        else:
            raise ValueError("unsupported VTU data array format %s" % fmt)
        return vals.reshape(-1, ncomp) if ncomp > 1 else vals

    pts_da = piece.find("./Points/DataArray")
    if pts_da is None:
        raise ValueError("VTU file has no Points array")
    out = dict(points=np.asarray(array_of(pts_da)).reshape(n_points, 3), cells=None, point_data={})
    pd = piece.find("./PointData")
    if pd is not None:
        for da in pd.findall("./DataArray"):
            a = np.asarray(array_of(da))
            out["point_data"][da.get("Name")] = a.astype(a.dtype.newbyteorder("="))
    return out



def _vtk_bytes(title, points, cells_flat, n_cells, cell_type, point_attributes):
    """The vtkio crate's legacy writer (Vtk::export_be, version 4.2), as used by vtk_format.rs:188-211."""
    pts = np.ascontiguousarray(points)
    ptype = "double" if pts.dtype == np.float64 else "float"
    b = bytearray()
    b += b"# vtk DataFile Version 4.2\n" + title.encode("ascii") + b"\nBINARY\n\nDATASET UNSTRUCTURED_GRID\n"
    b += ("POINTS %d %s\n" % (pts.shape[0], ptype)).encode("ascii")
    b += pts.astype(">f8" if ptype == "double" else ">f4").tobytes() + b"\n"
    b += ("\nCELLS %d %d\n" % (n_cells, cells_flat.size)).encode("ascii")
    b += cells_flat.astype(">i4").tobytes() + b"\n"
    b += ("\nCELL_TYPES %d\n" % n_cells).encode("ascii")
    b += np.full(n_cells, cell_type, dtype=">i4").tobytes() + b"\n"
    b += ("\nPOINT_DATA %d\n" % pts.shape[0]).encode("ascii")
    for name, data in (point_attributes or {}).items():
        a = np.ascontiguousarray(data)
        ncomp = 1 if a.ndim == 1 else a.shape[1]
        if a.dtype == np.uint64:
            ty, conv = "unsigned_long", ">u8"
        elif a.dtype == np.float64:
            ty, conv = "double", ">f8"
        else:
            ty, conv = "float", ">f4"
        b += ("\nSCALARS %s %s %d\nLOOKUP_TABLE default\n" % (name, ty, ncomp)).encode("ascii")
        b += a.astype(conv).tobytes() + b"\n"
    b += ("\nCELL_DATA %d\n\n" % n_cells).encode("ascii")
    return bytes(b)


# ------------------------------------------------------------------------------------------------------------
# PLY
# ------------------------------------------------------------------------------------------------------------
_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2", "int": "i4",
              "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def _read_ply(path):
    raw = open(path, "rb").read()
    end = raw.find(b"end_header")
    if not raw.startswith(b"ply") or end < 0:
        raise ValueError("not a PLY file")
    body = raw.find(b"\n", end) + 1
    fmt, elements = None, []
    for ln in raw[:end].decode("ascii", "replace").splitlines():
        t = ln.split()
        if not t:
            continue
        if t[0] == "format":
            fmt = t[1]
        elif t[0] == "element":
            elements.append(dict(name=t[1], count=int(t[2]), props=[]))
        elif t[0] == "property":
            if t[1] == "list":
                elements[-1]["props"].append(("list", t[2], t[3], t[4]))
            else:
                elements[-1]["props"].append(("scalar", t[1], t[2]))
    data = {}
    if fmt == "ascii":
        tokens = raw[body:].split()
        at = 0
        for el in elements:
            cols = {p[-1]: [] for p in el["props"]}
            for _ in range(el["count"]):
                for p in el["props"]:
                    if p[0] == "scalar":
                        cols[p[2]].append(float(tokens[at]))
                        at += 1
                    else:
                        k = int(tokens[at])
                        cols[p[3]].append([int(x) for x in tokens[at + 1:at + 1 + k]])
                        at += 1 + k
            data[el["name"]] = {p[-1]: (np.array(cols[p[-1]], dtype=_PLY_TYPES[p[1]]) if p[0] == "scalar" else cols[p[-1]]) for p in el["props"]}
        return data
    order = "<" if fmt == "binary_little_endian" else ">"
    at = body
    for el in elements:
        if all(p[0] == "scalar" for p in el["props"]):
            dt = np.dtype([(p[2], order + _PLY_TYPES[p[1]]) for p in el["props"]])
            arr = np.frombuffer(raw, dtype=dt, count=el["count"], offset=at)
            at += el["count"] * dt.itemsize
            data[el["name"]] = {p[2]: arr[p[2]].astype(arr[p[2]].dtype.newbyteorder("=")) for p in el["props"]}
        else:
            cols = {p[-1]: [] for p in el["props"]}
            # fast path: one list property with a constant count (triangles)
            if len(el["props"]) == 1 and el["count"] > 0:
                p = el["props"][0]
                cdt, idt = np.dtype(order + _PLY_TYPES[p[1]]), np.dtype(order + _PLY_TYPES[p[2]])
                k = int(np.frombuffer(raw, dtype=cdt, count=1, offset=at)[0])
                rec = np.dtype([("n", cdt), ("v", idt, (k,))])
                if at + el["count"] * rec.itemsize <= len(raw):
                    arr = np.frombuffer(raw, dtype=rec, count=el["count"], offset=at)
                    if np.all(arr["n"] == k):
                        at += el["count"] * rec.itemsize
                        data[el["name"]] = {p[3]: arr["v"].astype(np.int64)}
                        continue
            for _ in range(el["count"]):
                for p in el["props"]:
                    if p[0] == "scalar":
                        dt = np.dtype(order + _PLY_TYPES[p[1]])
                        cols[p[2]].append(np.frombuffer(raw, dtype=dt, count=1, offset=at)[0])
                        at += dt.itemsize
                    else:
                        cdt, idt = np.dtype(order + _PLY_TYPES[p[1]]), np.dtype(order + _PLY_TYPES[p[2]])
                        k = int(np.frombuffer(raw, dtype=cdt, count=1, offset=at)[0])
                        at += cdt.itemsize
                        cols[p[3]].append(np.frombuffer(raw, dtype=idt, count=k, offset=at).astype(np.int64).tolist())
                        at += k * idt.itemsize
            data[el["name"]] = {name: (np.array(v) if v and not isinstance(v[0], list) else v) for name, v in cols.items()}
    return data


# ------------------------------------------------------------------------------------------------------------
# BGEO (v5, particles)
# ------------------------------------------------------------------------------------------------------------
def _read_bgeo(path, want_attributes=False):
    """BGEO version 5 as the reference parses it (bgeo_format.rs:365-640): header, point attribute definitions, then per
    point x, y, z, an unnamed float ("unknown" in the reference) and the named attributes.  Returns (positions,
    {name: array}) with float attributes as float32, integer ones as uint64, 3-vectors as (N, 3) float32
    (bgeo_format.rs:332-350)."""
    raw = open(path, "rb").read()
    if raw[:2] == b"\x1f\x8b":
        raw = gzip.decompress(raw)
    if raw[:4] == b"\x7fNSJ":
        raise ValueError("unsupported BGEO format version (the new JSON-like BGEO format)")
    if raw[:5] != b"BgeoV" or struct.unpack(">i", raw[5:9])[0] != 5:
        raise ValueError("unsupported BGEO file (expected version 5)")
    n_points, _n_prims, _n_pg, _n_prg, n_pattr = struct.unpack(">5i", raw[9:29])
    off = 41
    psize = 4
    defs = []
    for _ in range(n_pattr):
        (ln,) = struct.unpack(">H", raw[off:off + 2])
        name = raw[off + 2:off + 2 + ln].decode("utf-8")
        off += 2 + ln
        (size,) = struct.unpack(">H", raw[off:off + 2])
        off += 2
        (atype,) = struct.unpack(">i", raw[off:off + 4])
        off += 4
        if atype not in (0, 1, 5):  # Float, Int, Vector; strings are rejected by the reference as well
            raise ValueError("unsupported BGEO attribute type %d of attribute '%s'" % (atype, name))
        off += 4 * size   # default values
        defs.append((name, size, atype, psize))
        psize += size
    data = np.frombuffer(raw, dtype=">f4", count=psize * n_points, offset=off).reshape(n_points, psize)
    positions = data[:, :3].astype(np.float32)
    attrs = {}
    if want_attributes:
        ints = np.frombuffer(raw, dtype=">i4", count=psize * n_points, offset=off).reshape(n_points, psize)
        for name, size, atype, col in defs:
            if atype == 1:
                a = ints[:, col:col + size]
                if a.size and int(a.min()) < 0:
                    raise ValueError('Failed to convert attribute "%s": failed to convert integer attribute' % name)
                attrs[name] = a.astype(np.uint64).reshape(n_points) if size == 1 else a.astype(np.uint64)
            elif atype == 0:
                a = data[:, col:col + size].astype(np.float32)
                attrs[name] = a.reshape(n_points) if size == 1 else a
            else:
                if size != 3:
                    attrs[name] = ValueError('Failed to convert attribute "%s": unsupported vector attribute size: %d' % (name, size))
                else:
                    attrs[name] = data[:, col:col + 3].astype(np.float32)
    return positions, attrs


def _read_bgeo_points(path):
    return _read_bgeo(path)[0]


def _bgeo_bytes(particles):
    """particles_to_bgeo (bgeo_format.rs:108-257): no named attributes, the unnamed float is 1.0, end bytes 00 ff."""
    p = np.ascontiguousarray(particles, dtype=np.float32).reshape(-1, 3)
    n = int(p.shape[0])
    if n > 2 ** 31 - 1:
        raise ValueError("number of particles (%d) is too large for bgeo format (max %d)" % (n, 2 ** 31 - 1))
    rec = np.empty((n, 4), dtype=">f4")
    rec[:, :3] = p
    rec[:, 3] = 1.0
    return b"Bgeo" + b"V" + struct.pack(">i", 5) + struct.pack(">8i", n, 0, 0, 0, 0, 0, 0, 0) + rec.tobytes() + b"\x00\xff"


# ------------------------------------------------------------------------------------------------------------
# particles
# ------------------------------------------------------------------------------------------------------------
def particles_from_file(path, dtype=np.float32):
    """`splashsurf_lib::io::particles_from_file` (io.rs:17-43): (N, 3) array of the requested Real type."""
    e = _ext(path)
    if e == "vtk":
        p = _read_vtk(path, points_only=True)["points"]
    elif e == "vtu":
        p = _read_vtu(path)["points"]
    elif e == "xyz":  # xyz_format.rs:10-36: native-endian f32 triples, trailing partial record ignored
        raw = np.fromfile(path, dtype=np.float32)
        p = raw[: (raw.size // 3) * 3].reshape(-1, 3)
    elif e == "ply":
        v = _read_ply(path).get("vertex")
        if v is None:
            raise ValueError("PLY file is missing a 'vertex' element")
        if any(v[k].dtype != np.float32 for k in ("x", "y", "z")):
            raise ValueError("Vertex properties have wrong PLY data type (expected float)")  # ply_format.rs:57-61
        p = np.stack([v["x"], v["y"], v["z"]], axis=1)
    elif e == "bgeo":
        p = _read_bgeo_points(path)
    elif e == "json":  # json_format.rs:21-57: array of [x, y, z]
        p = np.asarray(json.load(open(path)), dtype=np.float64).reshape(-1, 3)
    else:
        raise ValueError('Unsupported file format extension "%s" for reading particles' % e)
    return np.ascontiguousarray(p, dtype=dtype)


def particle_attributes_from_file(path, names):
    """Point attributes of a particle file as the binary reads them for `-a` (splashsurf/src/io.rs:68-190): legacy VTK
    point data or BGEO attributes.  Returns {name: array}; a missing name is an error."""
    e = _ext(path)
    if e == "vtk":
        data = _read_vtk(path)["point_data"]
    elif e == "vtu":
        data = _read_vtu(path)["point_data"]
    elif e == "bgeo":
        data = _read_bgeo(path, want_attributes=True)[1]
    else:
        raise ValueError('Unsupported file format extension "%s" for reading attributes (VTK and BGEO carry attributes)' % e)
    missing = [n for n in names if n not in data]
    if missing:
        raise ValueError('Missing attribute(s) "%s" in input file' % '", "'.join(missing))
    out = {}
    for n in names:
        if isinstance(data[n], Exception):
            raise data[n]
        out[n] = np.asarray(data[n])
    return out


def particles_to_file(particles, path):
    """vtk / json / bgeo as the reference's CLI writes them (splashsurf/src/io.rs:200-230), plus raw xyz."""
    p = np.ascontiguousarray(particles)
    e = _ext(path)
    if e == "vtk":  # vtk_format.rs:159-169: one VERTEX cell per particle
        n = p.shape[0]
        cells = np.empty((n, 2), dtype=np.int64)
        cells[:, 0] = 1
        cells[:, 1] = np.arange(n)
        data = _vtk_bytes("particles", p if p.dtype == np.float64 else p.astype(np.float32), cells.reshape(-1), n, 1, None)
    elif e == "json":  # json_format.rs:59-93: serde_json of Vec<[R; 3]> (f32 values widen to f64 exactly)
        rows = ["[" + ",".join(_fmt_json(v) for v in row) + "]" for row in p.astype(np.float64)]
        data = ("[" + ",".join(rows) + "]").encode("ascii")
    elif e == "xyz":
        data = p.astype("<f4").tobytes()
    elif e == "bgeo":  # the binary compresses by default (flate2 "fast"); the deflate stream is zlib's here, the content the same
        data = gzip.compress(_bgeo_bytes(p), compresslevel=1, mtime=0)
    else:
        raise ValueError('Unsupported file format extension "%s" for writing particles' % e)
    with open(path, "wb") as f:
        f.write(data)


# ------------------------------------------------------------------------------------------------------------
# meshes
# ------------------------------------------------------------------------------------------------------------
class MeshWithData:
    """vertices (V,3), triangles (T,3 uint64), point_attributes: ordered dict name -> (V,) or (V,3) array."""

    def __init__(self, vertices, triangles, point_attributes=None):
        self.vertices = np.ascontiguousarray(vertices)
        self.triangles = np.ascontiguousarray(triangles, dtype=np.uint64).reshape(-1, 3)
        self.point_attributes = dict(point_attributes or {})


def mesh_from_file(path, dtype=np.float32):
    """Surface meshes from vtk / ply (splashsurf/src/io.rs:245-262) and obj (obj_format.rs:73-190)."""
    e = _ext(path)
    if e == "vtk":
        d = _read_vtk(path)
        n_cells, flat = d["cells"] if d["cells"] is not None else (0, np.zeros(0, np.int64))
        flat = np.asarray(flat, dtype=np.int64)
        if flat.size and flat[0] == 0:  # "Sometimes VTK files from paraview start with an empty cell" (vtk_format.rs:271-273)
            flat = flat[1:]
        if flat.size % 4 != 0 or (flat.size and np.any(flat.reshape(-1, 4)[:, 0] != 3)):
            raise ValueError("Expected only triangle cells")
        tris = flat.reshape(-1, 4)[:, 1:]
        attrs = {k: np.ascontiguousarray(v, dtype=dtype) if v.dtype.kind == "f" else v for k, v in d["point_data"].items()}
        return MeshWithData(np.ascontiguousarray(d["points"], dtype=dtype), tris, attrs)
    if e == "ply":
        d = _read_ply(path)
        v = d.get("vertex")
        f = d.get("face")
        if v is None or f is None:
            raise ValueError("PLY file is missing a 'vertex' or 'face' element")
        verts = np.stack([v["x"], v["y"], v["z"]], axis=1).astype(dtype)
        idx = f.get("vertex_indices", f.get("vertex_index"))
        tris = np.asarray(idx, dtype=np.int64).reshape(-1, 3)
        attrs = {}
        names = [k for k in v if k not in ("x", "y", "z")]
        if all(k in v for k in ("nx", "ny", "nz")):
            attrs["normals"] = np.stack([v["nx"], v["ny"], v["nz"]], axis=1).astype(dtype)
        for k in names:
            if k in ("nx", "ny", "nz"):
                continue
            if k.endswith("_x") and k[:-2] + "_y" in v and k[:-2] + "_z" in v:
                attrs[k[:-2]] = np.stack([v[k], v[k[:-2] + "_y"], v[k[:-2] + "_z"]], axis=1).astype(dtype)
            elif k.endswith(("_y", "_z")) and k[:-2] + "_x" in v:
                continue
            else:
                attrs[k] = v[k].astype(dtype) if v[k].dtype.kind == "f" else v[k]
        # keep the file's property order (normals sit where nx was)
        ordered = {}
        for k in names:
            key = "normals" if k == "nx" else (k[:-2] if k.endswith("_x") and k[:-2] in attrs else k)
            if key in attrs and key not in ordered:
                ordered[key] = attrs[key]
        return MeshWithData(verts, tris, ordered)
    if e == "obj":
        verts, normals, tris = [], [], []
        for ln in open(path):
            t = ln.split()
            if not t:
                continue
            if t[0] == "v":
                verts.append([float(x) for x in t[1:4]])
            elif t[0] == "vn":
                normals.append([float(x) for x in t[1:4]])
            elif t[0] == "f":
                tris.append([int(x.split("/")[0]) - 1 for x in t[1:4]])
        if normals and len(normals) != len(verts):  # obj_format.rs:152-157 (the reference asserts)
            raise ValueError("length of vertex and vertex normal array doesn't match")
        attrs = {"normals": np.asarray(normals, dtype=dtype).reshape(-1, 3)} if normals else {}
        return MeshWithData(np.asarray(verts, dtype=dtype).reshape(-1, 3), np.asarray(tris, dtype=np.int64).reshape(-1, 3), attrs)
    raise ValueError('Unsupported file format extension "%s" for reading surface meshes' % e)


def mesh_to_file(mesh, path, point_attributes=None):
    """`mesh` is a MeshWithData or anything with `.vertices` / `.triangles` (optionally `.point_attributes`).
    vtk: vtk_format.rs:188-211 (title "mesh"); ply: ply_format.rs:190-268; obj: obj_format.rs:17-71."""
    v = np.ascontiguousarray(mesh.vertices)
    t = np.ascontiguousarray(mesh.triangles).astype(np.int64).reshape(-1, 3)
    attrs = dict(point_attributes if point_attributes is not None else getattr(mesh, "point_attributes", None) or {})
    e = _ext(path)
    if e == "vtk":
        cells = np.empty((t.shape[0], 4), dtype=np.int64)
        cells[:, 0] = 3
        cells[:, 1:] = t
        data = _vtk_bytes("mesh", v, cells.reshape(-1), t.shape[0], 5, attrs)
    elif e == "ply":
        hdr = ["ply", "format binary_little_endian 1.0", "element vertex %d" % v.shape[0], "property float x", "property float y", "property float z"]
        cols = [v[:, 0], v[:, 1], v[:, 2]]
        fields = [("x", "<f4"), ("y", "<f4"), ("z", "<f4")]
        for name, a in attrs.items():
            a = np.asarray(a)
            if name == "normals":
                hdr += ["property float nx", "property float ny", "property float nz"]
                comp = [("nx", a[:, 0]), ("ny", a[:, 1]), ("nz", a[:, 2])]
            elif a.dtype == np.uint64:
                hdr.append("property uint %s" % name)
                comp = [(name, a)]
            elif a.ndim == 1:
                hdr.append("property float %s" % name)
                comp = [(name, a)]
            else:
                hdr += ["property float %s_%s" % (name, c) for c in "xyz"]
                comp = [("%s_%s" % (name, c), a[:, i]) for i, c in enumerate("xyz")]
            for cname, col in comp:
                fields.append((cname, "<u4" if col.dtype == np.uint64 else "<f4"))
                cols.append(col)
        hdr += ["element face %d" % t.shape[0], "property list uchar uint vertex_indices", "end_header"]
        rec = np.empty(v.shape[0], dtype=np.dtype(fields))
        for (cname, _), col in zip(fields, cols):
            rec[cname] = col
        faces = np.empty(t.shape[0], dtype=np.dtype([("n", "u1"), ("v", "<u4", (3,))]))
        faces["n"] = 3
        faces["v"] = t
        data = ("\n".join(hdr) + "\n").encode("ascii") + rec.tobytes() + faces.tobytes()
    elif e == "obj":
        lines = ["v %s %s %s" % tuple(_fmt_display(x) for x in row) for row in v]
        normals = attrs.get("normals")
        if normals is not None:
            lines += ["vn %s %s %s" % tuple(_fmt_display(x) for x in row) for row in np.asarray(normals)]
            lines += ["f " + " ".join("%d//%d" % (i + 1, i + 1) for i in row) for row in t]
        else:
            lines += ["f " + " ".join("%d" % (i + 1) for i in row) for row in t]
        data = ("\n".join(lines) + "\n").encode("ascii") if lines else b""
    else:
        raise ValueError('Unsupported file format extension "%s" for writing surface meshes' % e)
    with open(path, "wb") as f:
        f.write(data)
