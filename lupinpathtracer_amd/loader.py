"""Scene and image I/O with the semantics of the reference's `lupin_loader` crate
(lupin_loader/src/loader.rs): the built-in Cornell box, Yocto/GL v2.4 JSON scenes with binary
little-endian PLY meshes and PNG / Radiance-HDR textures, and RGBE `.hdr` read/write.

Everything here is host-side preparation of the hot path's inputs; it produces an
`api.SceneCPU` + textures and hands them to `api.build_accel_structures_and_upload`.
"""
import json
import os
import struct
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import api
from ._abi import ENVIRONMENT_DTYPE, INSTANCE_DTYPE, MATERIAL_DTYPE, MESH_INFO_DTYPE, SENTINEL_IDX


@dataclass
class SceneCamera:  # loader.rs:303-308
    transform: np.ndarray = field(default_factory=api.identity_mat3x4)
    params: api.CameraParams = field(default_factory=api.CameraParams)


class LoadError(Exception):
    pass


# ------------------------------------------------------------------------------------------------
# Built-in Cornell box (build_scene_cornell_box, loader.rs:14-207; values from Yocto/GL)
# ------------------------------------------------------------------------------------------------

_BOX_INDICES = [0, 2, 1, 2, 0, 3, 4, 6, 5, 6, 4, 7, 8, 10, 9, 10, 8, 11, 12, 14, 13, 14, 12, 15,
                16, 18, 17, 18, 16, 19, 20, 22, 21, 22, 20, 23]

_CORNELL_MESHES = [
    # (name, vertices, indices, material)
    ("floor", [(-1, 0, 1), (1, 0, 1), (1, 0, -1), (-1, 0, -1)], [0, 1, 2, 2, 3, 0], 0),
    ("ceiling", [(-1, 2, 1), (-1, 2, -1), (1, 2, -1), (1, 2, 1)], [0, 1, 2, 2, 3, 0], 0),
    ("backwall", [(-1, 0, 1), (1, 0, 1), (1, 2, 1), (-1, 2, 1)], [0, 2, 1, 2, 0, 3], 0),
    ("rightwall", [(1, 0, -1), (1, 0, 1), (1, 2, 1), (1, 2, -1)], [0, 1, 2, 2, 3, 0], 2),
    ("leftwall", [(-1, 0, 1), (-1, 0, -1), (-1, 2, -1), (-1, 2, 1)], [0, 1, 2, 2, 3, 0], 1),
    ("shortbox", [(0.53, 0.6, -0.75), (0.7, 0.6, -0.17), (0.13, 0.6, -0.0), (-0.05, 0.6, -0.57), (-0.05, 0.0, -0.57),
                  (-0.05, 0.6, -0.57), (0.13, 0.6, -0.0), (0.13, 0.0, -0.0), (0.53, 0.0, -0.75), (0.53, 0.6, -0.75),
                  (-0.05, 0.6, -0.57), (-0.05, 0.0, -0.57), (0.7, 0.0, -0.17), (0.7, 0.6, -0.17), (0.53, 0.6, -0.75),
                  (0.53, 0.0, -0.75), (0.13, 0.0, -0.0), (0.13, 0.6, -0.0), (0.7, 0.6, -0.17), (0.7, 0.0, -0.17),
                  (0.53, 0.0, -0.75), (0.7, 0.0, -0.17), (0.13, 0.0, -0.0), (-0.05, 0.0, -0.57)], _BOX_INDICES, 0),
    ("tallbox", [(-0.53, 1.2, -0.09), (0.04, 1.2, 0.09), (-0.14, 1.2, 0.67), (-0.71, 1.2, 0.49), (-0.53, 0.0, -0.09),
                 (-0.53, 1.2, -0.09), (-0.71, 1.2, 0.49), (-0.71, 0.0, 0.49), (-0.71, 0.0, 0.49), (-0.71, 1.2, 0.49),
                 (-0.14, 1.2, 0.67), (-0.14, 0.0, 0.67), (-0.14, 0.0, 0.67), (-0.14, 1.2, 0.67), (0.04, 1.2, 0.09),
                 (0.04, 0.0, 0.09), (0.04, 0.0, 0.09), (0.04, 1.2, 0.09), (-0.53, 1.2, -0.09), (-0.53, 0.0, -0.09),
                 (-0.53, 0.0, -0.09), (0.04, 0.0, 0.09), (-0.14, 0.0, 0.67), (-0.71, 0.0, 0.49)], _BOX_INDICES, 0),
    ("light", [(-0.25, 1.99, -0.25), (-0.25, 1.99, 0.25), (0.25, 1.99, 0.25), (0.25, 1.99, -0.25)], [0, 2, 1, 2, 0, 3], 3),
]


def cornell_box_scene_cpu():
    """The SceneCPU of build_scene_cornell_box and its single camera."""
    scene = api.SceneCPU()
    mats = []
    for color, emission in (((0.725, 0.71, 0.68), None), ((0.63, 0.065, 0.05), None), ((0.14, 0.45, 0.091), None),
                            (None, (17.0, 12.0, 4.0))):
        m = api.default_material()
        if color is not None:
            m["color"] = (*color, 1.0)
        if emission is not None:
            m["emission"] = (*emission, 0.0)
        mats.append(m)
    scene.materials = np.array(mats, MATERIAL_DTYPE)
    infos, insts = [], []
    for mesh_idx, (_, verts, indices, mat) in enumerate(_CORNELL_MESHES):
        v = np.zeros((len(verts), 4), np.float32)
        v[:, :3] = np.array(verts, np.float32)
        scene.verts_pos_array.append(v)
        scene.indices_array.append(np.array(indices, np.uint32))
        infos.append(api.default_mesh_info())
        inst = api.default_instance()
        inst["mesh_idx"] = mesh_idx
        inst["mat_idx"] = mat
        insts.append(inst)
    scene.mesh_infos = np.array(infos, MESH_INFO_DTYPE)
    scene.instances = np.array(insts, INSTANCE_DTYPE)
    cam = SceneCamera(
        transform=np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [0, 1, -3.9]], np.float32),
        params=api.CameraParams(is_orthographic=False, lens=0.035, aperture=0.0, focus=3.9, film=0.024, aspect=1.0))
    return scene, [cam]


def build_scene_cornell_box(ctx, build_sw_and_hw=True):
    """lpl::build_scene_cornell_box (loader.rs:14-207) -> (Scene, [SceneCamera])"""
    scene_cpu, cams = cornell_box_scene_cpu()
    api.validate_scene(scene_cpu, 0, 0)
    return api.build_accel_structures_and_upload(ctx, scene_cpu, [], [], build_sw_and_hw), cams


def build_scene_empty(ctx):
    """lpl::build_scene_empty (loader.rs:7-12)"""
    scene_cpu = api.SceneCPU()
    api.validate_scene(scene_cpu, 0, 0)
    return api.build_accel_structures_and_upload(ctx, scene_cpu, [], [], True)


# ------------------------------------------------------------------------------------------------
# Images
# ------------------------------------------------------------------------------------------------

def read_hdr(path):
    """Radiance RGBE -> (H, W, 3) float32.  Decoding rule of the `image` crate the reference uses
    (loader.rs:218,1750): value = mantissa * 2^(e - 136), e == 0 -> 0."""
    with open(path, "rb") as f:
        data = f.read()
    pos = 0
    if not data.startswith(b"#?"):
        raise LoadError(f"{path}: not a Radiance file")
    # header lines until an empty line
    while True:
        end = data.index(b"\n", pos)
        line = data[pos:end]
        pos = end + 1
        if line.strip() == b"":
            break
    end = data.index(b"\n", pos)
    res = data[pos:end].split()
    pos = end + 1
    if len(res) != 4 or res[0] != b"-Y" or res[2] != b"+X":
        raise LoadError(f"{path}: unsupported orientation {res}")
    h, w = int(res[1]), int(res[3])
    buf = np.frombuffer(data, np.uint8)
    rgbe = np.zeros((h, w, 4), np.uint8)
    for y in range(h):
        if w < 8 or w > 0x7FFF or not (buf[pos] == 2 and buf[pos + 1] == 2 and (buf[pos + 2] & 0x80) == 0):
            # flat scanline
            rgbe[y] = buf[pos:pos + 4 * w].reshape(w, 4)
            pos += 4 * w
            continue
        if ((int(buf[pos + 2]) << 8) | int(buf[pos + 3])) != w:
            raise LoadError(f"{path}: bad scanline width")
        pos += 4
        for c in range(4):
            x = 0
            row = rgbe[y, :, c]
            while x < w:
                n = int(buf[pos])
                pos += 1
                if n > 128:
                    n -= 128
                    row[x:x + n] = buf[pos]
                    pos += 1
                else:
                    row[x:x + n] = buf[pos:pos + n]
                    pos += n
                x += n
    e = rgbe[..., 3].astype(np.int32)
    scale = np.where(e == 0, np.float32(0.0), np.exp2((e - 136).astype(np.float32))).astype(np.float32)
    return (rgbe[..., :3].astype(np.float32) * scale[..., None]).astype(np.float32)


def write_hdr(path, rgb):
    """(H, W, 3) float32 -> flat (un-RLE'd) Radiance file.  Pixel rule of the encoder the reference
    saves with (`image` HdrEncoder via save_texture, loader.rs:1775-1879)."""
    rgb = np.asarray(rgb, np.float32)
    h, w, _ = rgb.shape
    mx = rgb.max(axis=2)
    pos = mx > 0
    exp = np.zeros((h, w), np.int32)
    exp[pos] = np.floor(np.log2(mx[pos])).astype(np.int32) + 1
    mul = np.exp2(exp.astype(np.float32))
    mant = np.trunc(np.clip(rgb / mul[..., None] * 256.0, 0, 255)).astype(np.uint8)
    out = np.zeros((h, w, 4), np.uint8)
    out[..., :3] = np.where(pos[..., None], mant, 0)
    out[..., 3] = np.where(pos, np.clip(exp + 128, 0, 255), 0).astype(np.uint8)
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n")
        f.write(f"-Y {h} +X {w}\n".encode())
        f.write(out.tobytes())


def load_texture_pixels(path):
    """load_texture_with_usage (loader.rs:214-286): HDR/EXR -> Rgba16Float, everything else -> Rgba8Unorm."""
    ext = os.path.splitext(path)[1].lower().lstrip(".")
    if ext == "hdr":
        rgb = read_hdr(path)
        rgba = np.ones(rgb.shape[:2] + (4,), np.float32)
        rgba[..., :3] = rgb
        return rgba.astype(np.float16), rgba   # rgba32f_to_rgba16f (half::f16::from_f32 = round to nearest even)
    if ext == "exr":
        raise LoadError("EXR textures are not supported by this loader")
    from PIL import Image
    img = np.asarray(Image.open(path).convert("RGBA"), np.uint8)
    return np.ascontiguousarray(img), (img.astype(np.float32) / np.float32(255.0))


def save_texture(path, texture_or_array):
    """lpl::save_texture for `.hdr` (loader.rs:1775-1879; alpha dropped) -- accepts an api.Texture or (H,W,>=3)."""
    arr = texture_or_array.download() if hasattr(texture_or_array, "download") else np.asarray(texture_or_array)
    if arr.dtype == np.uint8:   # Rgba8Unorm (a tonemapped target): 8-bit RGB by file extension (loader.rs:1823-1851)
        from PIL import Image
        Image.fromarray(np.ascontiguousarray(arr[..., :3])).save(path)
        return
    write_hdr(path, arr[..., :3].astype(np.float32))


# ------------------------------------------------------------------------------------------------
# PLY (load_mesh_ply, loader.rs:1274-1566)
# ------------------------------------------------------------------------------------------------

def load_mesh_ply(path, scene: api.SceneCPU):
    with open(path, "rb") as f:
        data = f.read()
    end = data.find(b"end_header")
    if end < 0 or not data.startswith(b"ply"):
        raise LoadError(f"{path}: invalid PLY")
    header_end = data.index(b"\n", end) + 1
    lines = data[:header_end].decode("ascii", "replace").splitlines()
    num_verts = num_faces = 0
    offsets, offset = {}, 0
    section = None
    for line in lines[1:]:
        tok = line.split()
        if not tok or tok[0] == "comment":
            continue
        if tok[0] == "format":
            if tok[1] != "binary_little_endian" or tok[2] != "1.0":
                raise LoadError(f"{path}: only binary_little_endian 1.0 is supported")
        elif tok[0] == "element":
            section = tok[1]
            if section == "vertex":
                num_verts = int(tok[2])
            elif section == "face":
                num_faces = int(tok[2])
        elif tok[0] == "property":
            if section == "vertex":
                # only `float` properties occupy space in the reference's reader (loader.rs:1337-1342)
                size = 4 if tok[1] == "float" else 0
                name = {"s": "u", "t": "v"}.get(tok[2], tok[2])
                offsets[name] = offset
                offset += size
            elif section == "face":
                if tok[1:4] != ["list", "uchar", "uint"] and tok[1:4] != ["list", "uchar", "int"]:
                    raise LoadError(f"{path}: unsupported face property")
    stride = offset
    if not all(k in offsets for k in "xyz"):
        raise LoadError(f"{path}: missing positions")
    body = np.frombuffer(data, np.uint8, offset=header_end)
    vbytes = body[:num_verts * stride].reshape(num_verts, stride) if stride else np.zeros((num_verts, 0), np.uint8)

    def column(name):
        o = offsets[name]
        return np.ascontiguousarray(vbytes[:, o:o + 4]).view("<f4").reshape(-1)

    info = api.default_mesh_info()
    pos = np.zeros((num_verts, 4), np.float32)
    for i, k in enumerate("xyz"):
        pos[:, i] = column(k)
    if any(k in offsets for k in ("nx", "ny", "nz")):
        nrm = np.zeros((num_verts, 4), np.float32)
        for i, k in enumerate(("nx", "ny", "nz")):
            nrm[:, i] = column(k)
        scene.verts_normal_array.append(nrm)
        info["normals_buf_idx"] = len(scene.verts_normal_array) - 1
    if "u" in offsets or "v" in offsets:
        uv = np.zeros((num_verts, 2), np.float32)
        uv[:, 0] = column("u")
        uv[:, 1] = np.float32(1.0) - column("v")   # V flip (loader.rs:1431-1435)
        scene.verts_texcoord_array.append(uv)
        info["texcoords_buf_idx"] = len(scene.verts_texcoord_array) - 1
    if any(k in offsets for k in ("red", "green", "blue", "alpha")):
        col = np.zeros((num_verts, 4), np.float32)
        for i, k in enumerate(("red", "green", "blue", "alpha")):
            col[:, i] = column(k)
        scene.verts_color_array.append(col)
        info["colors_buf_idx"] = len(scene.verts_color_array) - 1

    # faces: uchar count + count * u32, fan-triangulated (ply_extract_indices, loader.rs:1535-1566)
    fbytes = body[num_verts * stride:]
    indices = None
    if num_faces and len(fbytes) >= num_faces * 13 and np.all(fbytes[0:num_faces * 13:13] == 3) and len(fbytes) < num_faces * 13 + 13:
        rec = fbytes[:num_faces * 13].reshape(num_faces, 13)
        indices = np.ascontiguousarray(rec[:, 1:]).view("<u4").reshape(-1).astype(np.uint32)
    else:
        out, p = [], 0
        raw = fbytes.tobytes()
        for _ in range(num_faces):
            n = raw[p]
            p += 1
            ids = struct.unpack_from(f"<{n}I", raw, p)
            p += 4 * n
            for j in range(1, n - 1):
                out += [ids[0], ids[j], ids[j + 1]]
        indices = np.array(out, np.uint32)
    if len(indices) and int(indices.max()) >= num_verts:
        raise LoadError(f"{path}: vertex index out of range")

    scene.mesh_infos = np.append(scene.mesh_infos, np.array([info], MESH_INFO_DTYPE))
    scene.verts_pos_array.append(pos)
    scene.indices_array.append(indices)
    return len(scene.mesh_infos) - 1


# ------------------------------------------------------------------------------------------------
# Yocto/GL v2.4 JSON (load_scene_yoctogl_v24, loader.rs:331-768; parse_material_yocto_v24, :770-911)
# ------------------------------------------------------------------------------------------------

_MAT_TYPES = {"matte": 0, "glossy": 1, "reflective": 2, "transparent": 3, "refractive": 4, "subsurface": 5,
              "volume": 6, "gltfpbr": 7}


def _mat3x4(frame):
    """parse_mat3x4f (loader.rs:1074-1097): 12 numbers, column by column -> (4 cols, 3 rows)."""
    return np.array(frame, np.float32).reshape(4, 3)


def _mul3x4(a, b):
    """Mat3x4 * Mat3x4 (base.rs:738-757), f32."""
    a4 = np.zeros((4, 4), np.float32)
    b4 = np.zeros((4, 4), np.float32)
    a4[:, :3], b4[:, :3] = a, b
    a4[3, 3] = b4[3, 3] = 1.0
    res = np.zeros((4, 3), np.float32)
    for i in range(3):
        for j in range(4):
            acc = np.float32(0.0)
            for k in range(4):
                acc = np.float32(acc + np.float32(a4[k][i] * b4[j][k]))
            res[j][i] = acc
    return res


def _parse_material(d):
    m = api.default_material()
    for key, val in d.items():   # file order matters: "color" resets opacity to 1 (loader.rs:791-792)
        if key == "color":
            m["color"] = (val[0], val[1], val[2], 1.0)
        elif key == "emission":
            m["emission"][:3] = val
        elif key == "scattering":
            m["scattering"][:3] = val
        elif key == "roughness":
            m["roughness"] = val
        elif key == "metallic":
            m["metallic"] = val
        elif key == "ior":
            m["ior"] = val
        elif key == "scanisotropy":
            m["sc_anisotropy"] = val
        elif key == "trdepth":
            m["tr_depth"] = val
        elif key == "opacity":
            m["color"][3] = val
        elif key == "type":
            if val in _MAT_TYPES:
                m["mat_type"] = _MAT_TYPES[val]
        elif key in ("color_tex", "emission_tex", "roughness_tex", "scattering_tex", "normal_tex"):
            m[key + "_idx"] = int(val) & 0xFFFFFFFF
    return m


def _find_asset(rel, dirs):
    for d in dirs:
        p = os.path.join(d, rel)
        if os.path.exists(p):
            return p
    raise LoadError(f"asset {rel} not found in {dirs}")


def load_scene_cpu_yoctogl_v24(path, asset_dirs: Sequence[str] = ()):
    """Parse a Yocto/GL 2.4 scene into (SceneCPU, textures, envs_info, cameras) without touching a device."""
    parent = os.path.dirname(os.path.abspath(path))
    dirs = [parent, *asset_dirs]
    with open(path, "r") as f:
        doc = json.load(f)   # dict order == file order

    conversion = api.identity_mat3x4()
    conversion[2][2] = -1.0   # Z flip into Lupin's left-handed frame (loader.rs:345-349)

    scene = api.SceneCPU()
    cams: List[SceneCamera] = []
    tex_paths: List[Optional[str]] = []
    tex_referenced = 0

    def note_tex(idx):
        nonlocal tex_referenced
        if idx != SENTINEL_IDX:
            tex_referenced = max(tex_referenced, idx + 1)

    for section, items in doc.items():
        if section == "cameras":
            for c in items:
                cam = SceneCamera()
                for key, val in c.items():
                    if key == "aspect":
                        cam.params.aspect = float(val)
                    elif key == "focus":
                        cam.params.focus = float(val)
                    elif key == "aperture":
                        cam.params.aperture = float(val)
                    elif key == "lens":
                        cam.params.lens = float(val)
                    elif key == "film":
                        cam.params.film = float(val)
                    elif key == "orthographic":
                        cam.params.is_orthographic = bool(val)
                    elif key == "frame":
                        cam.transform = _mul3x4(_mul3x4(conversion, _mat3x4(val)), conversion)
                cams.append(cam)
        elif section == "environments":
            env = api.default_environment()
            env["transform"][2][2] = -1.0   # conversion_mat4 * IDENTITY
            envs = []
            for e in items:   # NOTE: `env` is declared outside the loop in the reference: fields persist (loader.rs:444-445)
                for key, val in e.items():
                    if key == "emission":
                        env["emission"] = val
                    elif key == "emission_tex":
                        env["emission_tex_idx"] = int(val) & 0xFFFFFFFF
                        note_tex(int(env["emission_tex_idx"]))
                    elif key == "frame":
                        fm = _mat3x4(val)
                        t = np.zeros((4, 4), np.float32)
                        t[:, :3] = fm
                        t[3, 3] = 1.0
                        t[:, 2] *= np.float32(-1.0)   # conversion_mat4 * m: negates row 2 (z) of every column
                        env["transform"] = t
                envs.append(env.copy())
            scene.environments = np.array(envs, ENVIRONMENT_DTYPE)
        elif section == "textures":
            for t in items:
                uri = t.get("uri", "")
                tex_paths.append(uri if uri else None)
        elif section == "materials":
            mats = [_parse_material(m) for m in items]
            for m in mats:
                for k in ("color_tex_idx", "emission_tex_idx", "roughness_tex_idx", "scattering_tex_idx", "normal_tex_idx"):
                    note_tex(int(m[k]))
            scene.materials = np.array(mats, MATERIAL_DTYPE)
        elif section == "shapes":
            for s in items:
                uri = s.get("uri", "")
                if uri:
                    if not uri.lower().endswith(".ply"):
                        raise LoadError(f"unsupported shape format: {uri}")
                    load_mesh_ply(_find_asset(uri, dirs), scene)
        elif section == "instances":
            insts = []
            for it in items:
                transform = _mul3x4(conversion, api.identity_mat3x4())
                mesh_idx = mat_idx = 0
                for key, val in it.items():
                    if key == "frame":
                        transform = _mul3x4(conversion, _mat3x4(val))
                    elif key == "material":
                        mat_idx = int(val)
                    elif key == "shape":
                        mesh_idx = int(val)
                insts.append(api.instance_from_transform(transform, mesh_idx, mat_idx))
            scene.instances = np.array(insts, INSTANCE_DTYPE)

    n_tex = max(len(tex_paths), tex_referenced)
    tex_paths += [None] * (n_tex - len(tex_paths))
    textures, tex_f32 = [], []
    for p in tex_paths:
        if p is None:
            raise LoadError("texture referenced but not declared")
        px, f32 = load_texture_pixels(_find_asset(p, dirs))
        textures.append(api.TextureCPU(px))
        tex_f32.append(f32)

    envs_info = []
    for env in scene.environments:
        ti = int(env["emission_tex_idx"])
        if ti == SENTINEL_IDX:
            envs_info.append(api.EnvMapInfo(np.ones((1, 1, 4), np.float32), 1, 1))   # loader.rs:728-737
        else:
            f = tex_f32[ti]
            envs_info.append(api.EnvMapInfo(np.ascontiguousarray(f, np.float32), f.shape[1], f.shape[0]))

    api.validate_scene(scene, len(textures), len(textures))
    return scene, textures, envs_info, cams


def load_scene_yoctogl_v24(path, ctx, build_both_bvhs=True, asset_dirs: Sequence[str] = (), blas_builder="sah"):
    """lpl::load_scene_yoctogl_v24 (loader.rs:331) -> (Scene, [SceneCamera])"""
    scene_cpu, textures, envs_info, cams = load_scene_cpu_yoctogl_v24(path, asset_dirs)
    return api.build_accel_structures_and_upload(ctx, scene_cpu, textures, envs_info, build_both_bvhs, blas_builder=blas_builder), cams


def build_scene_bistro_class_cpu(asset_dir, seed=0xB157, n_meshes=20, n_instances=400, n_lights=100, n_materials=60):
    """Seeded procedural stand-in for the 'bistroexterior' config (the real scene is not
    available offline): `n_meshes` jittered copies of bunny.ply (144 046 triangles each, ~2.9 M unique triangles at
    the default), instanced `n_instances` times on a street-like grid with random scale / yaw, `n_materials`
    materials cycling through every material type, `n_lights` small emissive quads (street lamps / windows), a
    textured ground quad and the sky.hdr environment.  Deterministic in `seed`.
    Returns (SceneCPU, textures, envs_info, cameras)."""
    rng = np.random.default_rng(seed)
    scene = api.SceneCPU()
    base = api.SceneCPU()
    load_mesh_ply(os.path.join(asset_dir, "shapes", "bunny.ply"), base)
    bpos, bidx = base.verts_pos_array[0], base.indices_array[0]
    bnrm = base.verts_normal_array[0] if base.verts_normal_array else None
    lo, hi = bpos[:, :3].min(0), bpos[:, :3].max(0)
    centre, extent = (lo + hi) / 2, float((hi - lo).max())

    infos = []
    for m in range(n_meshes):
        # low-frequency warp: every copy is a different mesh with its own BVH
        k = rng.uniform(2.0, 9.0, 3) / extent
        ph = rng.uniform(0, 2 * np.pi, 3)
        amp = rng.uniform(0.01, 0.04) * extent
        p = bpos.copy()
        q = bpos[:, :3] - centre
        p[:, 0] += (amp * np.sin(k[0] * q[:, 1] + ph[0])).astype(np.float32)
        p[:, 1] += (amp * np.sin(k[1] * q[:, 2] + ph[1])).astype(np.float32)
        p[:, 2] += (amp * np.sin(k[2] * q[:, 0] + ph[2])).astype(np.float32)
        scene.verts_pos_array.append(p.astype(np.float32))
        scene.indices_array.append(bidx.copy())
        info = api.default_mesh_info()
        if bnrm is not None and m % 2 == 0:   # half the meshes shade with vertex normals, half with geometric ones
            scene.verts_normal_array.append(bnrm.copy())
            info["normals_buf_idx"] = len(scene.verts_normal_array) - 1
        infos.append(info)
    # ground quad (textured) and a unit light quad
    ground = len(infos)
    g = np.zeros((4, 4), np.float32)
    g[:, :3] = [(-1, 0, 1), (1, 0, 1), (1, 0, -1), (-1, 0, -1)]
    scene.verts_pos_array.append(g)
    scene.indices_array.append(np.array([0, 1, 2, 2, 3, 0], np.uint32))
    scene.verts_texcoord_array.append(np.array([(0, 0), (40, 0), (40, 40), (0, 40)], np.float32))
    gi = api.default_mesh_info()
    gi["texcoords_buf_idx"] = 0
    infos.append(gi)
    quad = len(infos)
    qv = np.zeros((4, 4), np.float32)
    qv[:, :3] = [(-0.5, 0, -0.5), (-0.5, 0, 0.5), (0.5, 0, 0.5), (0.5, 0, -0.5)]
    scene.verts_pos_array.append(qv)
    scene.indices_array.append(np.array([0, 2, 1, 2, 0, 3], np.uint32))
    infos.append(api.default_mesh_info())
    scene.mesh_infos = np.array(infos, MESH_INFO_DTYPE)

    mats = []
    for i in range(n_materials):
        m = api.default_material()
        t = i % 8
        m["mat_type"] = t
        m["color"] = (*rng.uniform(0.25, 0.95, 3), 1.0)
        m["roughness"] = [0.0, 0.2, 0.15, 0.0, 0.0, 0.1, 0.0, 0.3][t] + (rng.uniform(0, 0.2) if t in (1, 2, 7) else 0.0)
        m["metallic"] = rng.uniform(0, 1) if t == 7 else 0.0
        m["ior"] = 1.5
        if t in (4, 5, 6):
            m["scattering"][:3] = rng.uniform(0.1, 0.9, 3)
            m["tr_depth"] = 0.05
        mats.append(m)
    ground_mat = len(mats)
    gm = api.default_material()
    gm["color"] = (0.7, 0.7, 0.7, 1.0)
    gm["color_tex_idx"] = 0
    mats.append(gm)
    light_mats = []
    for i in range(8):
        lm = api.default_material()
        lm["emission"][:3] = rng.uniform(8.0, 30.0) * np.array([1.0, rng.uniform(0.7, 1.0), rng.uniform(0.4, 1.0)])
        light_mats.append(len(mats))
        mats.append(lm)
    scene.materials = np.array(mats, MATERIAL_DTYPE)

    def frame(scale, yaw, pos):
        c, s_ = np.cos(yaw), np.sin(yaw)
        f = np.zeros((4, 3), np.float32)
        f[0] = (scale * c, 0, -scale * s_)
        f[1] = (0, scale, 0)
        f[2] = (scale * s_, 0, scale * c)
        f[3] = pos
        return f

    insts = []
    side = int(np.ceil(np.sqrt(n_instances)))
    spacing = 1.6
    unit = 1.0 / extent
    for i in range(n_instances):
        gx, gz = i % side, i // side
        sc_ = unit * rng.uniform(0.6, 1.4)
        pos = np.array([(gx - side / 2) * spacing + rng.uniform(-0.3, 0.3), 0.0, gz * spacing + rng.uniform(-0.3, 0.3) + 2.0])
        pos[1] = -lo[1] * sc_
        fr = frame(sc_, rng.uniform(0, 2 * np.pi), pos - centre * np.array([sc_, 0, sc_]) * np.array([1, 0, 1]))
        insts.append(api.instance_from_transform(fr, int(rng.integers(0, n_meshes)), int(rng.integers(0, n_materials))))
    half = side * spacing
    insts.append(api.instance_from_transform(frame(half, 0.0, np.array([0.0, 0.0, half * 0.5 + 1.0])), ground, ground_mat))
    for i in range(n_lights):
        pos = np.array([rng.uniform(-half / 2, half / 2), rng.uniform(1.5, 3.0), rng.uniform(2.0, side * spacing)])
        insts.append(api.instance_from_transform(frame(rng.uniform(0.15, 0.4), rng.uniform(0, 2 * np.pi), pos), quad,
                                                 light_mats[int(rng.integers(0, len(light_mats)))]))
    scene.instances = np.array(insts, INSTANCE_DTYPE)

    env = api.default_environment()
    env["emission"] = (0.35, 0.35, 0.4)
    env["emission_tex_idx"] = 1
    scene.environments = np.array([env], ENVIRONMENT_DTYPE)
    floor_px, _ = load_texture_pixels(os.path.join(asset_dir, "textures", "floor.png"))
    sky_px, sky_f32 = load_texture_pixels(os.path.join(asset_dir, "textures", "sky.hdr"))
    textures = [api.TextureCPU(floor_px), api.TextureCPU(sky_px)]
    envs_info = [api.EnvMapInfo(np.ascontiguousarray(sky_f32, np.float32), sky_f32.shape[1], sky_f32.shape[0])]
    cam = SceneCamera(transform=np.array([[1, 0, 0], [0, 0.94, 0.34], [0, -0.34, 0.94], [0, 2.6, -3.0]], np.float32),
                      params=api.CameraParams(lens=0.035, film=0.036, aspect=16 / 9, focus=10000.0, aperture=0.0))
    api.validate_scene(scene, len(textures), len(textures))
    return scene, textures, envs_info, [cam]


def build_scene_bistro_class(ctx, asset_dir, blas_builder="sah", **kw):
    scene_cpu, textures, envs_info, cams = build_scene_bistro_class_cpu(asset_dir, **kw)
    return api.build_accel_structures_and_upload(ctx, scene_cpu, textures, envs_info, True, blas_builder=blas_builder), cams


def compute_dimensions_for_1080p(aspect):
    """lupin_tests/src/main.rs:477-484"""
    if aspect < 1.0:
        return int(np.float32(1920.0) * np.float32(aspect)), 1920
    return 1920, int(np.float32(1920.0) / np.float32(aspect))
