"""Generate tests/golden/camera_reference.npz by IMPORTING the reference's own helpers
(/root/reference/Articulation/utils/nerf2physic_utils.py) in the dev container.  The output is
data only (inputs + the reference's outputs); no reference source is stored.  Re-run with:
    python tests/golden/make_camera_golden.py
Needs /root/reference; the tests that consume the .npz do not.
"""
import importlib.util
import json
import os
import struct
import tempfile

import numpy as np

REF = "/root/reference/Articulation"
HERE = os.path.dirname(os.path.abspath(__file__))

spec = importlib.util.spec_from_file_location("ref_n2p", os.path.join(REF, "utils", "nerf2physic_utils.py"))
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

out = {}
rng = np.random.default_rng(2025)

# --- 1. project_3d_to_2d on the six committed look-at cameras (camera_params.json) -----------
with open(os.path.join(REF, "openbox_output", "segmentation", "camera_params.json")) as f:
    cams = json.load(f)
names = sorted(cams)
out["cam_names"] = np.array(names)
out["cam_c2w"] = np.stack([np.array(cams[k]["c2w"], dtype=np.float64) for k in names])
out["cam_K"] = np.stack([np.array(cams[k]["intrinsics"], dtype=np.float64) for k in names])
out["cam_res"] = np.stack([np.array(cams[k]["resolution"], dtype=np.int64) for k in names])
pts = rng.uniform(-2.0, 2.0, size=(64, 3))
out["pts"] = pts
uv, dist = [], []
for i, k in enumerate(names):
    w2c = np.linalg.inv(out["cam_c2w"][i])
    a, b = ref.project_3d_to_2d(pts, w2c, out["cam_K"][i], return_dists=True)
    uv.append(a)
    dist.append(b)
out["proj_uv"] = np.stack(uv)
out["proj_dist"] = np.stack(dist)

# --- 2. depth <-> distance, unproject ----------------------------------------------------------
K = np.array([[310.0, 0, 47.5], [0, 305.0, 30.25], [0, 0, 1]])
depth = rng.uniform(0.5, 6.0, size=(61, 96))
out["dd_K"] = K
out["dd_depth"] = depth
out["dd_distance"] = ref.depth_to_distance(depth, K)
out["dd_roundtrip"] = ref.distance_to_depth(out["dd_distance"], K)
c2w = out["cam_c2w"][2]
px = np.array([[3, 5], [90, 60], [47, 30]])
out["unproj_px"] = px
out["unproj_c2w"] = c2w
out["unproj_xyz"] = np.stack([ref.unproject_point((int(u), int(v)), depth, c2w, K) for u, v in px])

# --- 3. transforms.json / dataparser_transforms.json parsing ----------------------------------
frames = []
for i in range(3):
    m = np.eye(4)
    m[:3, :3] = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    m[:3, 3] = rng.normal(size=3)
    frames.append({"file_path": f"images/frame_{i:05d}.png", "transform_matrix": m.tolist(),
                   "fl_x": 500.0 + i, "fl_y": 501.0 + i, "cx": 320.0, "cy": 240.0})
tj = {"fl_x": 600.0, "fl_y": 601.0, "cx": 400.0, "cy": 300.0, "w": 800, "h": 600, "frames": frames}
dp = {"transform": (np.hstack([np.linalg.qr(rng.normal(size=(3, 3)))[0], rng.normal(size=(3, 1))])).tolist(),
      "scale": 0.37}
out["transforms_json"] = np.array(json.dumps(tj))
out["dataparser_json"] = np.array(json.dumps(dp))
with tempfile.TemporaryDirectory() as d:
    pt, pd = os.path.join(d, "transforms.json"), os.path.join(d, "dataparser_transforms.json")
    json.dump(tj, open(pt, "w"))
    json.dump(dp, open(pd, "w"))
    c2ws, Kg = ref.parse_transforms_json(pt)
    w2cs, Ks = ref.parse_transforms_json(pt, return_w2c=True, different_Ks=True)
    T, s = ref.parse_dataparser_transforms_json(pd)
out["tj_c2ws"], out["tj_K_global"] = np.stack(c2ws), Kg
out["tj_w2cs"], out["tj_K_frames"] = np.stack(w2cs), np.stack(Ks)
out["dp_transform"], out["dp_scale"] = T, np.float64(s)
# the un-normalisation load_ns_point_cloud applies (nerf2physic_utils.py:68-74), on seeded points
T4 = np.concatenate([T, np.array([[0, 0, 0, 1 / s]])], 0)
inv = np.linalg.inv(T4)
p_ns = rng.normal(size=(16, 3))
ph = np.concatenate([p_ns, np.ones((16, 1))], 1) @ inv.T
out["dp_points_ns"] = p_ns
out["dp_points_world"] = ph[:, :3] / ph[:, 3:]

# --- 4. real-data pin of the projection convention: part-mesh face centroids vs SAM masks -----
def glb_centroids(path):
    with open(path, "rb") as f:
        data = f.read()
    magic, version, length = struct.unpack_from("<4sII", data, 0)
    assert magic == b"glTF"
    off, chunks = 12, []
    while off < length:
        clen, ctype = struct.unpack_from("<II", data, off)
        chunks.append((ctype, data[off + 8: off + 8 + clen]))
        off += 8 + clen
    gltf = json.loads(chunks[0][1])
    binc = chunks[1][1]

    def acc(i):
        a = gltf["accessors"][i]
        bv = gltf["bufferViews"][a["bufferView"]]
        dt = {5126: np.float32, 5125: np.uint32, 5123: np.uint16, 5121: np.uint8}[a["componentType"]]
        nc = {"SCALAR": 1, "VEC3": 3}[a["type"]]
        start = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
        return np.frombuffer(binc, dtype=dt, count=a["count"] * nc, offset=start).reshape(a["count"], nc)
    cents = []
    for node in gltf["nodes"]:
        if "mesh" not in node:
            continue
        M = np.array(node.get("matrix", np.eye(4).T.reshape(-1).tolist())).reshape(4, 4).T
        for prim in gltf["meshes"][node["mesh"]]["primitives"]:
            v = acc(prim["attributes"]["POSITION"]).astype(np.float64)
            v = v @ M[:3, :3].T + M[:3, 3]
            idx = acc(prim["indices"]).reshape(-1, 3).astype(np.int64)
            cents.append(v[idx].mean(axis=1))
    return np.concatenate(cents)

parts = os.path.join(REF, "openbox_output", "parts")
seg = os.path.join(REF, "openbox_output", "segmentation")
lid, body = glb_centroids(os.path.join(parts, "lid.glb")), glb_centroids(os.path.join(parts, "body.glb"))
# every 8th centroid keeps the fixture small and still covers both masks densely
out["lid_centroids"] = lid[::8].astype(np.float32)
out["body_centroids"] = body[::8].astype(np.float32)
for colour in ("RED", "GREEN"):
    m = np.load(os.path.join(seg, f"mask_{colour}_bottom.npy"))
    out[f"mask_{colour}_shape"] = np.array(m.shape)
    out[f"mask_{colour}_bits"] = np.packbits(m.astype(bool))
out["mask_camera"] = np.array("bottom")

np.savez_compressed(os.path.join(HERE, "camera_reference.npz"), **out)
print("wrote camera_reference.npz:", {k: getattr(v, "shape", None) for k, v in out.items()})
