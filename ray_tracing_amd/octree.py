"""Host-side construction of the linearised voxel octree handed to the C ABI.

The reference builds `FastOctTree<Voxel>` by repeated `set` calls with leaf collapsing
(cpu_rt/src/ray_tracer/hittable/fast_oct_tree/operations/set.rs:4-36, mod.rs:43-169) and loads
MagicaVoxel files through dot_vox (operations/load.rs:74-127). Construction stays on the host; this
module produces the same tree bottom-up from the final voxel set with numpy (a node whose eight
children are equal leaves is one leaf, untouched regions are maximal empty nodes -- except the chain of
small empty cells that growing the world leaves in the origin corner, `growth_spine` -- and the world
edge is the smallest power of two above the largest coordinate ever set) and flattens it into the
`mrt_octree` arrays. Only the query side (`trace_ray`, `ray_iteration`) runs on the GPU.
"""
from __future__ import annotations

import ctypes as C
import struct
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import _ffi as F
from .scene import VoxelGrid

NODE_DTYPE = np.dtype([("kind", "<i4"), ("size", "<u4"), ("first_child", "<i4"), ("voxel", "<i4")])


def from_color_hex(s: str) -> Tuple[float, float, float]:
    """RgbColor::from_color_hex (cpu_rt/src/prelude/rgb_color.rs:92-126)."""
    if not s.startswith("#") or len(s) < 7:
        raise ValueError("invalid syntax")
    total = int(s[1:7], 16)
    f = np.float32
    return (float(f((total >> 16) & 0xFF) / f(255.0)), float(f((total >> 8) & 0xFF) / f(255.0)),
            float(f(total & 0xFF) / f(255.0)))


# Voxel records (fast_oct_tree/voxel.rs:12-45) as rows of VOXEL_ROW float64 values:
# kind, albedo r g b, fuzz, edge_kind, edge_probability, edge_fuzz, edge albedo r g b, density
VOXEL_ROW = 12


def solid_lambertian(albedo) -> np.ndarray:
    """Voxel::Solid(SolidVoxel::Lambertian { albedo })"""
    r = np.zeros(VOXEL_ROW)
    r[0], r[1:4] = F.VOX_LAMBERTIAN, np.float32(albedo)
    return r


def solid_reflect(albedo, fuzz: float) -> np.ndarray:
    """Voxel::Solid(SolidVoxel::Reflect { albedo, fuzz })"""
    r = np.zeros(VOXEL_ROW)
    r[0], r[1:4], r[4] = F.VOX_REFLECT, np.float32(albedo), np.float32(fuzz)
    return r


def volume(density: float, color, edge_probability: float = 0.0, edge_material: Optional[np.ndarray] = None) -> np.ndarray:
    """Voxel::Volume(VolumeVoxel { density, color, edge_effect }); edge_material = a solid_lambertian /
    solid_reflect row for VolumeEdgeEffect::Solid { hit_probability, solid_material }, None for ::None."""
    r = np.zeros(VOXEL_ROW)
    r[0], r[1:4], r[11] = F.VOX_VOLUME, np.float32(color), float(density)
    if edge_material is not None:
        r[5] = F.VOX_EDGE_REFLECT if edge_material[0] == F.VOX_REFLECT else F.VOX_EDGE_LAMBERTIAN
        r[6], r[7], r[8:11] = np.float32(edge_probability), edge_material[4], edge_material[1:4]
    return r


class OctTree:
    """A set of voxels; later `set` calls overwrite earlier ones."""

    def __init__(self):
        self._coords: List[np.ndarray] = []
        self._rows: List[np.ndarray] = []
        self._max = -1

    def set_rows(self, coords: np.ndarray, rows: np.ndarray) -> None:
        """FastOctTree::set for every coordinate; `rows`: one voxel record or one per coordinate."""
        coords = np.asarray(coords, dtype=np.int64).reshape(-1, 3)
        if len(coords) == 0:
            return
        if coords.min() < 0:
            raise ValueError("negative voxel coordinate")
        rows = np.asarray(rows, dtype=np.float64)
        if rows.ndim == 1:
            rows = np.broadcast_to(rows, (len(coords), VOXEL_ROW))
        self._coords.append(coords)
        self._rows.append(np.ascontiguousarray(rows, dtype=np.float64))
        self._max = max(self._max, int(coords.max()))

    def set_many(self, coords: np.ndarray, albedo) -> None:
        """Solid Lambertian voxels with the given albedo (one colour or one per coordinate)."""
        coords = np.asarray(coords, dtype=np.int64).reshape(-1, 3)
        col = np.asarray(albedo, dtype=np.float32)
        if col.ndim == 1:
            col = np.broadcast_to(col, (len(coords), 3))
        rows = np.zeros((len(coords), VOXEL_ROW))
        rows[:, 0], rows[:, 1:4] = F.VOX_LAMBERTIAN, col
        self.set_rows(coords, rows)

    def set(self, albedo, position) -> None:
        self.set_many(np.asarray([position]), albedo)

    def voxels(self) -> Tuple[np.ndarray, np.ndarray]:
        """Final (coords, albedo / colour) with overwrites resolved (last set wins)."""
        c, rows = self.voxel_rows()
        return c, np.ascontiguousarray(rows[:, 1:4], dtype=np.float32)

    def voxel_rows(self) -> Tuple[np.ndarray, np.ndarray]:
        """Final (coords, voxel records) with overwrites resolved (last set wins)."""
        if not self._coords:
            return np.zeros((0, 3), np.int64), np.zeros((0, VOXEL_ROW))
        c = np.concatenate(self._coords)
        col = np.concatenate(self._rows)
        key = (c[:, 0] << 42) | (c[:, 1] << 21) | c[:, 2]
        # keep the LAST occurrence of every key
        order = np.argsort(key[::-1], kind="stable")
        rk = key[::-1][order]
        first = np.ones(len(rk), dtype=bool)
        first[1:] = rk[1:] != rk[:-1]
        idx = (len(key) - 1 - order)[first]
        return c[idx], col[idx]

    def combine(self, other: "OctTree", offset) -> "OctTree":
        """FastOctTree::combine (operations/combine.rs:4-19): other's voxels are set at +offset;
        positions that become negative are dropped."""
        c, rows = other.voxel_rows()
        c = c + np.asarray(offset, dtype=np.int64)
        keep = (c >= 0).all(axis=1)
        self.set_rows(c[keep], rows[keep])
        return self

    def world_log2(self) -> Optional[int]:
        if self._max < 0:
            return None
        k = 0
        while (1 << k) <= self._max:
            k += 1
        return k

    def world_size(self) -> int:
        k = self.world_log2()
        return 0 if k is None else 1 << k

    def linearise(self) -> VoxelGrid:
        """Bottom-up canonical octree -> (nodes, children, voxel table, root)."""
        coords, rows = self.voxel_rows()
        k = self.world_log2()
        if k is None:
            raise ValueError("empty octree has no bounding box (hittable.rs:69-79 returns None)")
        # distinct voxels by exact equality (the reference's VolumeVoxel equality has a 1e-4 margin,
        # voxel.rs:42-50, which only matters for voxels that differ by less than that)
        records, ids = np.unique(rows, axis=0, return_inverse=True)
        ids = np.asarray(ids).reshape(-1)
        voxel_table = np.ascontiguousarray(records[:, 1:4], dtype=np.float32)
        LEAF, PARENT, EMPTY = F.OCT_LEAF, F.OCT_PARENT, F.OCT_EMPTY
        out_kind: List[np.ndarray] = []
        out_size: List[np.ndarray] = []
        out_fc: List[np.ndarray] = []
        out_vox: List[np.ndarray] = []
        children_blocks: List[np.ndarray] = []
        n_nodes = 0
        n_child_blocks = 0
        # current level: one entry per occupied cell
        cx, cy, cz = coords[:, 0].copy(), coords[:, 1].copy(), coords[:, 2].copy()
        kind = np.full(len(coords), LEAF, dtype=np.int32)
        payload = ids.astype(np.int64)          # leaf: voxel id; parent: first_child offset
        for level in range(1, k + 1):
            key = ((cx >> 1) << 42) | ((cy >> 1) << 21) | (cz >> 1)
            slot = ((cx & 1) << 2) | ((cy & 1) << 1) | (cz & 1)
            order = np.argsort(key, kind="stable")
            key, slot, kind, payload = key[order], slot[order], kind[order], payload[order]
            ukey, start = np.unique(key, return_index=True)
            parent_of = np.searchsorted(ukey, key)
            npar = len(ukey)
            ck = np.full((npar, 8), EMPTY, dtype=np.int32)
            cp = np.full((npar, 8), -1, dtype=np.int64)
            ck[parent_of, slot] = kind
            cp[parent_of, slot] = payload
            collapse = (ck == LEAF).all(axis=1) & (cp == cp[:, :1]).all(axis=1)
            keep = ~collapse
            nk = int(keep.sum())
            # emit the 8 child records of every surviving parent
            ek, ep = ck[keep].reshape(-1), cp[keep].reshape(-1)
            out_kind.append(ek)
            out_size.append(np.full(len(ek), level - 1, dtype=np.uint32))
            out_fc.append(np.where(ek == PARENT, ep, -1).astype(np.int32))
            out_vox.append(np.where(ek == LEAF, ep, -1).astype(np.int32))
            children_blocks.append((n_nodes + np.arange(len(ek))).astype(np.uint32))
            first_child = (n_child_blocks + np.arange(nk)) * 8
            n_nodes += len(ek)
            n_child_blocks += nk
            new_kind = np.where(collapse, LEAF, PARENT).astype(np.int32)
            new_payload = cp[:, 0].copy()
            new_payload[keep] = first_child
            cx, cy, cz = ukey >> 42, (ukey >> 21) & ((1 << 21) - 1), ukey & ((1 << 21) - 1)
            kind, payload = new_kind, new_payload
        assert len(kind) == 1 and cx[0] == 0 and cy[0] == 0 and cz[0] == 0
        out_kind.append(kind.astype(np.int32))
        out_size.append(np.array([k], dtype=np.uint32))
        out_fc.append(np.where(kind == PARENT, payload, -1).astype(np.int32))
        out_vox.append(np.where(kind == LEAF, payload, -1).astype(np.int32))
        nodes = np.zeros(n_nodes + 1, dtype=NODE_DTYPE)
        nodes["kind"] = np.concatenate(out_kind)
        nodes["size"] = np.concatenate(out_size)
        nodes["first_child"] = np.concatenate(out_fc)
        nodes["voxel"] = np.concatenate(out_vox)
        children = np.concatenate(children_blocks) if children_blocks else np.zeros(0, np.uint32)
        nodes, children = growth_spine(nodes, children.astype(np.uint32), n_nodes)
        return VoxelGrid(nodes=nodes, children=children, voxels=voxel_table, root=n_nodes, records=records)


def growth_spine(nodes: np.ndarray, children: np.ndarray, root: int):
    """The one place where the tree the reference's `set` sequence ends with is not the canonical tree of the voxel
    set. `set` grows the world by wrapping the old root as child 0 of a new parent with seven empty siblings
    (operations/set.rs:11-30), starting from an empty root of size 0, and `set_child` re-collapses nodes only along
    the path of the voxel being set (mod.rs:129-165). When the first voxel ever set is not at the origin, the wrapped
    empty roots therefore stay behind as a chain of parents with eight empty children in the origin corner: on the
    path root -> child 0 -> child 0 ..., every node that holds no voxel is a parent of eight empty children (down to
    size 1) where the canonical tree has one large empty node. A ray through that corner steps through the small
    empty cells. Appends those parents' children to (nodes, children); the root keeps its index."""
    nodes, children = nodes.copy(), children.copy()
    cur = root
    while True:
        kind, size = int(nodes[cur]["kind"]), int(nodes[cur]["size"])
        if kind == F.OCT_PARENT:
            cur = int(children[int(nodes[cur]["first_child"])])   # child 0
            continue
        if kind != F.OCT_EMPTY or size < 1:
            return nodes, children
        kids = np.zeros(8, dtype=NODE_DTYPE)
        kids["kind"], kids["size"], kids["first_child"], kids["voxel"] = F.OCT_EMPTY, size - 1, -1, -1
        nodes[cur]["kind"], nodes[cur]["first_child"] = F.OCT_PARENT, len(children)
        children = np.concatenate([children, (len(nodes) + np.arange(8)).astype(np.uint32)])
        nodes = np.concatenate([nodes, kids])


# ------------------------------------------------------------------ MagicaVoxel .vox (dot_vox 5.1.1 semantics)
@dataclass
class VoxFile:
    models: List[np.ndarray]   # each (n, 4) uint8: x, y, z, i  (i = file colour index - 1)
    palette: np.ndarray        # (256, 4) uint8 RGBA, entry i = i-th entry of the RGBA chunk


def parse_vox(data: bytes) -> VoxFile:
    if data[:4] != b"VOX ":
        raise ValueError("not a .vox file")
    pos = 8
    models: List[np.ndarray] = []
    palette = None

    def read_chunk(p):
        cid = data[p:p + 4]
        n, m = struct.unpack_from("<ii", data, p + 4)
        return cid, p + 12, n, m

    cid, body, n, m = read_chunk(pos)
    if cid != b"MAIN":
        raise ValueError("missing MAIN chunk")
    p, end = body + n, body + n + m
    while p < end:
        cid, body, n, m = read_chunk(p)
        if cid == b"XYZI":
            cnt = struct.unpack_from("<i", data, body)[0]
            v = np.frombuffer(data, dtype=np.uint8, count=cnt * 4, offset=body + 4).reshape(-1, 4).copy()
            v[:, 3] = v[:, 3] - 1     # dot_vox stores the file index minus one
            models.append(v)
        elif cid == b"RGBA":
            palette = np.frombuffer(data, dtype=np.uint8, count=1024, offset=body).reshape(256, 4).copy()
        p = body + n + m
    if palette is None:
        raise ValueError(".vox without an RGBA chunk (default palette) is not supported")
    return VoxFile(models, palette)


def load_vox(vox: VoxFile, tree_cls=None):
    """FastOctTree::load_vox (operations/load.rs:74-127): all models overlaid, scene-graph transforms
    ignored, x = vx - min_x, y = vz - min_z, z = (max_y - min_y) - (vy - min_y). `tree_cls`: OctTree
    (default) or NativeOctTree."""
    allv = np.concatenate(vox.models).astype(np.int64)
    mn, mx = allv[:, :3].min(axis=0), allv[:, :3].max(axis=0)
    tree = (tree_cls or OctTree)()
    for m in vox.models:
        v = m.astype(np.int64)
        pos = np.stack([v[:, 0] - mn[0], v[:, 2] - mn[2], -(v[:, 1] - mn[1]) + mx[1] - mn[1]], axis=1)
        rgb = vox.palette[v[:, 3], :3].astype(np.float32) / np.float32(255.0)
        tree.set_many(pos, rgb)
    return tree


def load_map(tiles, tile_size: int, models, tree_cls=None):
    """FastOctTree::load_map (operations/load.rs:28-67): `tiles[x][z]` names a tile type; `models` maps a type to
    its VoxFile (types without a model are air). Tile (x, z) is combined at ((len - x - 1) * tile_size, 0,
    z * tile_size), in x-major order, later tiles overwriting earlier ones."""
    trees = {k: load_vox(v, tree_cls) for k, v in models.items()}
    tree = (tree_cls or OctTree)()
    for x in range(len(tiles)):
        for z in range(len(tiles[x])):
            tile = trees.get(int(tiles[x][z]))
            if tile is not None:
                tree.combine(tile, ((len(tiles) - x - 1) * tile_size, 0, z * tile_size))
    return tree


def box_points(start, end) -> np.ndarray:
    """IterBox::from_xyz(end).start_xyz(start) (cpu_rt/src/prelude.rs:19-77): x-major, z innermost."""
    # BoxIter::next (prelude.rs:57-77) stops on x alone; y and z emit their start value even when the
    # end is not above it (e.g. IterBox::from_xyz(40, 0, 40) is the 40 x 1 x 40 layer y = 0)
    xs = np.arange(start[0], end[0])
    ys = np.arange(start[1], max(end[1], start[1] + 1))
    zs = np.arange(start[2], max(end[2], start[2] + 1))
    if len(xs) == 0 or len(ys) == 0 or len(zs) == 0:
        return np.zeros((0, 3), np.int64)
    g = np.stack(np.meshgrid(xs, ys, zs, indexing="ij"), axis=-1).reshape(-1, 3)
    return g.astype(np.int64)


def city_grid(building, tree_cls=None):
    """Geometry of `fast_oct_tree::city()` (cpu_rt/src/ray_tracer/world/fast_oct_tree.rs:818-883)."""
    ROAD_WIDTH, ROAD_LENGTH, LINE_POSITION, CELL_SIZE, NUMBER_GRIDS = 9, 400, 4, 64, 5
    line, road = from_color_hex("#e4af00"), from_color_hex("#363636")
    grid = (tree_cls or type(building))()
    for i in range(NUMBER_GRIDS):
        x_offset = i * (CELL_SIZE + ROAD_WIDTH)
        pts = box_points((x_offset, 0, 0), (ROAD_WIDTH + x_offset, 1, ROAD_LENGTH))
        col = np.where((pts[:, 0] == LINE_POSITION)[:, None], np.float32(line), np.float32(road))
        grid.set_many(pts, col)
        pts = box_points((0, 0, x_offset), (ROAD_LENGTH, 1, 20 + x_offset))
        col = np.where((pts[:, 2] == LINE_POSITION)[:, None], np.float32(line), np.float32(road))
        grid.set_many(pts, col)
    return grid.combine(building, (ROAD_WIDTH, 1, ROAD_WIDTH))


# ------------------------------------------------------------------ native builder (C ABI)
VOXEL_DTYPE = np.dtype([("kind", "<i4"), ("albedo", "<f4", 3), ("fuzz", "<f4"), ("edge_kind", "<i4"),
                        ("edge_probability", "<f4"), ("edge_fuzz", "<f4"), ("edge_albedo", "<f4", 3), ("reserved", "<i4"),
                        ("density", "<f8")])


def rows_to_voxels(rows: np.ndarray) -> np.ndarray:
    """VOXEL_ROW rows -> mrt_voxel records."""
    rows = np.asarray(rows, dtype=np.float64).reshape(-1, VOXEL_ROW)
    v = np.zeros(len(rows), dtype=VOXEL_DTYPE)
    v["kind"], v["albedo"], v["fuzz"] = rows[:, 0], rows[:, 1:4], rows[:, 4]
    v["edge_kind"], v["edge_probability"], v["edge_fuzz"] = rows[:, 5], rows[:, 6], rows[:, 7]
    v["edge_albedo"], v["density"] = rows[:, 8:11], rows[:, 11]
    return v


class NativeOctTree:
    """The same interface as OctTree on top of the library's builder (`mrt_voxels_*`, csrc/mrt_octree.cpp):
    what a compiled host would call. `linearise()` returns a VoxelGrid."""

    def __init__(self):
        self._lib = F.load()
        self._h = self._lib.mrt_voxels_create()
        if not self._h:
            raise MemoryError("mrt_voxels_create failed")

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.mrt_voxels_destroy(self._h)
            self._h = None

    def set_rows(self, coords: np.ndarray, rows: np.ndarray) -> None:
        xyz = np.ascontiguousarray(coords, dtype=np.int32).reshape(-1, 3)
        rows = np.asarray(rows, dtype=np.float64)
        one = rows.ndim == 1
        vox = rows_to_voxels(rows)
        F.check(self._lib.mrt_voxels_set(self._h, xyz.ctypes.data, vox.ctypes.data, len(xyz), 1 if one else 0))

    def set_many(self, coords: np.ndarray, albedo) -> None:
        coords = np.asarray(coords).reshape(-1, 3)
        col = np.asarray(albedo, dtype=np.float32)
        if col.ndim == 1:
            self.set_rows(coords, solid_lambertian(col))
        else:
            rows = np.zeros((len(coords), VOXEL_ROW))
            rows[:, 0], rows[:, 1:4] = F.VOX_LAMBERTIAN, col
            self.set_rows(coords, rows)

    def combine(self, other: "NativeOctTree", offset) -> "NativeOctTree":
        off = (C.c_int32 * 3)(*[int(v) for v in offset])
        F.check(self._lib.mrt_voxels_combine(self._h, other._h, off))
        return self

    def load_vox_bytes(self, data: bytes) -> "NativeOctTree":
        buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
        F.check(self._lib.mrt_voxels_load_vox(self._h, buf, len(data)))
        return self

    def __len__(self) -> int:
        return int(self._lib.mrt_voxels_count(self._h))

    def linearise(self) -> VoxelGrid:
        o = F.Octree()
        F.check(self._lib.mrt_voxels_linearise(self._h, C.byref(o)))
        nodes = np.ctypeslib.as_array(C.cast(o.nodes, C.POINTER(C.c_int32)), shape=(o.n_nodes, 4)).copy().view(NODE_DTYPE).reshape(-1)
        children = (np.ctypeslib.as_array(o.children, shape=(o.n_children,)).copy() if o.n_children
                    else np.zeros(0, np.uint32))
        raw = np.ctypeslib.as_array(C.cast(o.voxels, C.POINTER(C.c_uint8)), shape=(o.n_voxels * VOXEL_DTYPE.itemsize,)).copy()
        vox = raw.view(VOXEL_DTYPE)
        records = np.zeros((len(vox), VOXEL_ROW))
        records[:, 0], records[:, 1:4], records[:, 4] = vox["kind"], vox["albedo"], vox["fuzz"]
        records[:, 5], records[:, 6], records[:, 7] = vox["edge_kind"], vox["edge_probability"], vox["edge_fuzz"]
        records[:, 8:11], records[:, 11] = vox["edge_albedo"], vox["density"]
        return VoxelGrid(nodes=nodes, children=children.astype(np.uint32), voxels=np.ascontiguousarray(vox["albedo"]),
                         root=int(o.root), records=records)


def write_vox(models, palette: np.ndarray) -> bytes:
    """A minimal MagicaVoxel file (MAIN > SIZE/XYZI per model, RGBA) from decoded models; the inverse of
    parse_vox for what load_vox reads. `models`: arrays (n, 4) of x, y, z, i with i = file index - 1."""
    def chunk(cid: bytes, body: bytes, kids: bytes = b"") -> bytes:
        return cid + struct.pack("<ii", len(body), len(kids)) + body + kids
    kids = b""
    for m in models:
        m = np.asarray(m, dtype=np.uint8).reshape(-1, 4).copy()
        m[:, 3] = m[:, 3] + 1
        kids += chunk(b"SIZE", struct.pack("<iii", 256, 256, 256))
        kids += chunk(b"XYZI", struct.pack("<i", len(m)) + m.tobytes())
    kids += chunk(b"RGBA", np.asarray(palette, dtype=np.uint8).reshape(256, 4).tobytes())
    return b"VOX " + struct.pack("<i", 150) + chunk(b"MAIN", b"", kids)
