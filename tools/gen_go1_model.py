#!/usr/bin/env python3
"""Bake the Go1 model DATA (URDF numbers + actuator-net weights) into C headers.

Run in the authoring container only (needs /root/reference); the generated
headers are committed because /root/reference does not exist on the GPU box.

Sources (data only, no code is taken from the reference):
  resources/robots/go1/urdf/go1.urdf            links/joints/collision shapes
  resources/actuator_nets/unitree_go1.pt        6->32->32->1 softsign MLP weights
Asset options that shape the simulated tree (go1_gym/envs/base/legged_robot_config.py:220-241,
legged_robot.py:1494-1507): collapse_fixed_joints=True (base+trunk+imu_link merge,
thigh_shoulder links vanish), feet kept as separate bodies (urdf:188 dont_collapse),
replace_cylinder_with_capsule=True.  For the dynamics we additionally merge each
foot into its calf (a fixed joint is dynamically a composite body); the foot is
still reported as its own body (index 4,8,12,16) for positions/velocities/forces.

Body/DoF order is FL, FR, RL, RR (SURVEY App. A / go1_gym_deploy/envs/lcm_agent.py:64-68).
"""
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np

REF = os.environ.get("WTW_REFERENCE", "/root/reference")
URDF = os.path.join(REF, "resources/robots/go1/urdf/go1.urdf")
ACT = os.path.join(REF, "resources/actuator_nets/unitree_go1.pt")
OUT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "walk-these-ways_amd", "csrc")

LEGS = ["FL", "FR", "RL", "RR"]


def vec(s):
    return np.array([float(x) for x in s.split()])


def inertia_mat(el):
    g = lambda k: float(el.get(k))
    return np.array([[g("ixx"), g("ixy"), g("ixz")],
                     [g("ixy"), g("iyy"), g("iyz")],
                     [g("ixz"), g("iyz"), g("izz")]])


def link_inertial(root, name):
    link = root.find(f"./link[@name='{name}']")
    ine = link.find("inertial")
    o = ine.find("origin")
    com = vec(o.get("xyz")) if o is not None else np.zeros(3)
    if o is not None:
        assert np.allclose(vec(o.get("rpy")), 0)
    m = float(ine.find("mass").get("value"))
    I = inertia_mat(ine.find("inertia"))
    return m, com, I


def merge(parts):
    """parts: list of (m, com, Icom) expressed in one frame -> composite (m, com, Icom)."""
    M = sum(p[0] for p in parts)
    c = sum(p[0] * p[1] for p in parts) / M
    I = np.zeros((3, 3))
    for m, ci, Ii in parts:
        d = ci - c
        I += Ii + m * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
    return M, c, I


def joint(root, name):
    j = root.find(f"./joint[@name='{name}']")
    o = j.find("origin")
    assert np.allclose(vec(o.get("rpy")), 0)
    out = {"xyz": vec(o.get("xyz"))}
    ax = j.find("axis")
    if ax is not None:
        out["axis"] = vec(ax.get("xyz"))
    lim = j.find("limit")
    if lim is not None:
        out.update(lower=float(lim.get("lower")), upper=float(lim.get("upper")),
                   effort=float(lim.get("effort")), velocity=float(lim.get("velocity")))
    return out


def fmt(a, ctype="double"):
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        return "{" + ", ".join(f"{x:.17g}" for x in a) + "}"
    return "{\n" + ",\n".join("  " + fmt(r) for r in a) + "\n}"


def main():
    root = ET.parse(URDF).getroot()

    # ---- base: trunk + imu_link (fixed) -------------------------------------------------
    mt, ct, It = link_inertial(root, "trunk")
    mi, ci, Ii = link_inertial(root, "imu_link")
    ci = joint(root, "imu_joint")["xyz"] + ci
    mb, cb, Ib = merge([(mt, ct, It), (mi, ci, Ii)])

    masses = [mb]
    coms = [cb]
    inertias = [Ib]
    j_origin, j_axis, j_lo, j_hi, j_vel, j_eff = [], [], [], [], [], []
    foot_off = []
    hip_cap_center, hip_cap_half, hip_cap_r = [], None, None
    thigh_half = calf_half = None
    thigh_c = calf_c = None
    for leg in LEGS:
        for part in ["hip", "thigh", "calf"]:
            m, c, I = link_inertial(root, f"{leg}_{part}")
            if part == "calf":
                mf, cf, If = link_inertial(root, f"{leg}_foot")
                off = joint(root, f"{leg}_foot_fixed")["xyz"]
                m, c, I = merge([(m, c, I), (mf, off + cf, If)])
                foot_off.append(off)
            masses.append(m)
            coms.append(c)
            inertias.append(I)
            j = joint(root, f"{leg}_{part}_joint")
            j_origin.append(j["xyz"])
            ax = j["axis"]
            assert np.allclose(np.abs(ax).sum(), 1.0) and ax.max() == 1.0
            j_axis.append(int(np.argmax(ax)))
            j_lo.append(j["lower"]); j_hi.append(j["upper"])
            j_vel.append(j["velocity"]); j_eff.append(j["effort"])
            # collision shapes
            col = root.find(f"./link[@name='{leg}_{part}']/collision")
            o = col.find("origin")
            xyz, rpy = vec(o.get("xyz")), vec(o.get("rpy"))
            geom = col.find("geometry")
            if part == "hip":
                cyl = geom.find("cylinder")
                assert np.allclose(rpy, [np.pi / 2, 0, 0], atol=1e-6)   # axis -> body y
                hip_cap_center.append(xyz)
                hip_cap_half = float(cyl.get("length")) / 2
                hip_cap_r = float(cyl.get("radius"))
            else:
                box = vec(geom.find("box").get("size"))
                assert np.allclose(rpy, [0, np.pi / 2, 0], atol=1e-6)   # box x -> -body z, box z -> body x
                half = np.array([box[2], box[1], box[0]]) / 2
                if part == "thigh":
                    thigh_half, thigh_c = half, xyz
                else:
                    calf_half, calf_c = half, xyz
    trunk_box = vec(root.find("./link[@name='trunk']/collision/geometry/box").get("size")) / 2
    foot_r = float(root.find("./link[@name='FL_foot']/collision/geometry/sphere").get("radius"))

    total = sum(masses)
    assert abs(total - 11.309932) < 1e-5, total

    lines = []
    A = lines.append
    A("/* GENERATED by tools/gen_go1_model.py from the reference DATA files")
    A(" *   resources/robots/go1/urdf/go1.urdf (trunk :45-63, imu :65-79, legs :90-582)")
    A(" * Do not edit.  Body order: 0 base, then per leg (FL,FR,RL,RR): hip, thigh, calf(+foot).")
    A(" * Inertias are about each body's COM in body axes: xx,xy,xz,yy,yz,zz.  */")
    A("#ifndef GO1_MODEL_DATA_H")
    A("#define GO1_MODEL_DATA_H")
    A("#ifndef GO1_CONST")
    A("#define GO1_CONST static const")
    A("#endif")
    A("#ifndef GO1_REAL")
    A("#define GO1_REAL double   /* the HIP translation unit defines it as float */")
    A("#endif")
    A("#define GO1_NB 13      /* dynamic bodies */")
    A("#define GO1_NJ 12")
    A("#define GO1_NBODY_REPORT 17  /* base + 4*(hip,thigh,calf,foot): Isaac Gym body indexing */")
    A(f"#define GO1_TOTAL_MASS {total:.17g}")
    A(f"GO1_CONST GO1_REAL GO1_BODY_MASS[13] = {fmt(masses)};")
    A(f"GO1_CONST GO1_REAL GO1_BODY_COM[13][3] = {fmt(coms)};")
    I6 = [[I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2]] for I in inertias]
    A(f"GO1_CONST GO1_REAL GO1_BODY_INERTIA[13][6] = {fmt(I6)};")
    A(f"GO1_CONST GO1_REAL GO1_JOINT_ORIGIN[12][3] = {fmt(j_origin)};")
    A("GO1_CONST int GO1_JOINT_AXIS[12] = {" + ", ".join(map(str, j_axis)) + "};  /* 0=x 1=y */")
    A(f"GO1_CONST GO1_REAL GO1_JOINT_LOWER[12] = {fmt(j_lo)};")
    A(f"GO1_CONST GO1_REAL GO1_JOINT_UPPER[12] = {fmt(j_hi)};")
    A(f"GO1_CONST GO1_REAL GO1_JOINT_VEL_LIMIT[12] = {fmt(j_vel)};")
    A(f"GO1_CONST GO1_REAL GO1_JOINT_EFFORT[12] = {fmt(j_eff)};")
    A(f"GO1_CONST GO1_REAL GO1_FOOT_OFFSET[4][3] = {fmt(foot_off)};  /* in calf frame */")
    A(f"#define GO1_FOOT_RADIUS {foot_r:.17g}")
    A(f"GO1_CONST GO1_REAL GO1_TRUNK_BOX_HALF[3] = {fmt(trunk_box)};")
    A(f"GO1_CONST GO1_REAL GO1_HIP_CAPSULE_CENTER[4][3] = {fmt(hip_cap_center)};  /* axis = body y */")
    A(f"#define GO1_HIP_CAPSULE_HALF {hip_cap_half:.17g}")
    A(f"#define GO1_HIP_CAPSULE_RADIUS {hip_cap_r:.17g}")
    A(f"GO1_CONST GO1_REAL GO1_THIGH_BOX_HALF[3] = {fmt(thigh_half)};")
    A(f"GO1_CONST GO1_REAL GO1_THIGH_BOX_CENTER[3] = {fmt(thigh_c)};")
    A(f"GO1_CONST GO1_REAL GO1_CALF_BOX_HALF[3] = {fmt(calf_half)};")
    A(f"GO1_CONST GO1_REAL GO1_CALF_BOX_CENTER[3] = {fmt(calf_c)};")
    A("#endif")
    os.makedirs(OUT_DIR, exist_ok=True)
    with open(os.path.join(OUT_DIR, "go1_model_data.h"), "w") as f:
        f.write("\n".join(lines) + "\n")

    # ---- actuator net -------------------------------------------------------------------
    import torch
    net = torch.jit.load(ACT, map_location="cpu")
    sd = {k: v.detach().double().numpy() for k, v in net.named_parameters()}
    lines = []
    A = lines.append
    A("/* GENERATED by tools/gen_go1_model.py from resources/actuator_nets/unitree_go1.pt")
    A(" * (TorchScript Sequential: Linear(6,32) softsign Linear(32,32) softsign Linear(32,1);")
    A(" * loaded at go1_gym/envs/base/legged_robot.py:1238-1253).  fp32 weights printed exactly. */")
    A("#ifndef GO1_ACTUATOR_DATA_H")
    A("#define GO1_ACTUATOR_DATA_H")
    A("#ifndef GO1_CONST")
    A("#define GO1_CONST static const")
    A("#endif")
    A(f"GO1_CONST float GO1_ACT_W0[32][6] = {fmt(sd['0.weight'])};")
    A(f"GO1_CONST float GO1_ACT_B0[32] = {fmt(sd['0.bias'])};")
    A(f"GO1_CONST float GO1_ACT_W1[32][32] = {fmt(sd['2.weight'])};")
    A(f"GO1_CONST float GO1_ACT_B1[32] = {fmt(sd['2.bias'])};")
    A(f"GO1_CONST float GO1_ACT_W2[32] = {fmt(sd['4.weight'][0])};")
    A(f"GO1_CONST float GO1_ACT_B2 = {float(sd['4.bias'][0]):.17g};")
    A("#endif")
    with open(os.path.join(OUT_DIR, "go1_actuator_data.h"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("wrote", OUT_DIR, "total mass", total)


if __name__ == "__main__":
    sys.exit(main())
