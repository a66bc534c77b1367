"""Robot topology: URDF -> kinematic tree + standard inertial parameters.

This is the host-side replacement for what the reference obtains from iDynTree's
``ModelLoader`` / ``Model`` objects (``identification/model.py:60-68,112-131,190-192``)
and from ``helpers.URDFHelpers.getJointLimits/getJointFriction``
(``identification/helpers.py:898-973``).  It is own code; the rules it follows are
stated in SURVEY.md Appendix A and pinned by ``tests/test_topology.py`` against the
documented known answers (KUKA a-priori table, link counts 3/8/9/48, ...).

Conventions
-----------
* *Fake links* (massless links with exactly one neighbour that is attached by a fixed
  joint) are removed and survive as named frames rigidly attached to their neighbour.
* The base link is the URDF root, or its only child when the root itself is fake.
* **Serialisation** (which link owns column block ``10 l``, which joint owns DOF ``d``):
  ``order="traversal"`` (default) is the order in which a depth-first walk from the root
  visits the tree when the neighbours of a link are pushed on a LIFO stack in the document
  order of their joints and a link is numbered when it is popped -- the walk iDynTree's
  ``Model::computeFullTreeTraversal`` performs and from which its loader renumbers links
  and joints.  It reproduces every joint list the reference holds
  (``model/*_regressor.xml``, ``configs/walkman_static.yaml:60-64``: left leg, right leg,
  waist, left arm, right arm for WALK-MAN); ``tests/golden/reference_joint_orders.json``.
  ``order="document"`` keeps URDF document order of links / movable joints (rounds 1-2).
  Explicit name lists (``joint_names``, ``link_names``) override either -- the way to
  adopt whatever ``Model.jointNames`` / ``Model.linkNames`` an iDynTree build prints.
* Per-link standard parameters ``[m, m*cx, m*cy, m*cz, Ixx, Ixy, Ixz, Iyy, Iyz, Izz]``
  with the inertia expressed about the link-frame origin in link axes
  (``identification/model.py:220-231``).
* Joint rest transform: child frame pose in the parent frame, ``R = Rz(y) Ry(p) Rx(r)``,
  first ``<origin>`` element wins; joint axis is given in the child frame, normalised.
"""
from __future__ import annotations

import json
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import Any

import numpy as np

JOINT_FIXED = 0
JOINT_REVOLUTE = 1
JOINT_PRISMATIC = 2


def rpy_to_matrix(rpy) -> np.ndarray:
    """R = Rz(yaw) @ Ry(pitch) @ Rx(roll) (URDF / iDynTree ``Rotation::RPY``)."""
    r, p, y = (float(v) for v in rpy)
    cr, sr = np.cos(r), np.sin(r)
    cp, sp = np.cos(p), np.sin(p)
    cy, sy = np.cos(y), np.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def _vec(text: str | None, default) -> np.ndarray:
    if text is None:
        return np.array(default, dtype=float)
    return np.array([float(v) for v in text.split()], dtype=float)


def _origin(elem) -> tuple[np.ndarray, np.ndarray]:
    """(R, p) of the FIRST <origin> child of elem; identity if absent."""
    o = elem.find("origin") if elem is not None else None
    if o is None:
        return np.eye(3), np.zeros(3)
    xyz = _vec(o.attrib.get("xyz"), [0, 0, 0])
    rpy = _vec(o.attrib.get("rpy"), [0, 0, 0])
    return rpy_to_matrix(rpy), xyz


def inertial_to_params(mass: float, com: np.ndarray, I_com: np.ndarray) -> np.ndarray:
    """10 standard parameters from mass, COM (link frame) and inertia about the COM in link axes."""
    c = np.asarray(com, dtype=float)
    I_o = np.asarray(I_com, dtype=float) + mass * (float(c @ c) * np.eye(3) - np.outer(c, c))
    return np.array(
        [mass, mass * c[0], mass * c[1], mass * c[2], I_o[0, 0], I_o[0, 1], I_o[0, 2], I_o[1, 1], I_o[1, 2], I_o[2, 2]]
    )


@dataclass
class Topology:
    """Kinematic tree in the layout the C-ABI consumes (see include/fbr.h ``fbr_topology``)."""

    name: str
    link_names: list[str]
    parent: list[int]  # parent link index, -1 for the base
    joint_names: list[str]  # name of the joint connecting link to its parent ("" for the base)
    joint_type: list[int]  # JOINT_FIXED / JOINT_REVOLUTE / JOINT_PRISMATIC
    dof_index: list[int]  # DOF index of that joint, -1 when fixed / base
    rest_R: np.ndarray  # (L,3,3) child frame orientation in the parent frame at q=0
    rest_p: np.ndarray  # (L,3)   child frame origin in the parent frame
    axis: np.ndarray  # (L,3) unit joint axis in the child frame (zeros for fixed)
    params: np.ndarray  # (L,10) a-priori standard parameters
    dof_names: list[str]
    limits: dict[str, dict[str, float]] = field(default_factory=dict)
    friction: dict[str, dict[str, float]] = field(default_factory=dict)
    frames: dict[str, dict[str, Any]] = field(default_factory=dict)  # name -> {link, R(3x3), p(3)}
    # URDF document ranks, kept so that either serialisation can be rebuilt from a stored topology:
    doc_link_rank: list[int] = field(default_factory=list)  # rank of the <link> element among the kept links
    doc_joint_rank: list[int] = field(default_factory=list)  # rank of the link's parent <joint> element (-1: base)

    # ------------------------------------------------------------------ sizes
    @property
    def num_links(self) -> int:
        return len(self.link_names)

    @property
    def num_dofs(self) -> int:
        return len(self.dof_names)

    @property
    def base_index(self) -> int:
        return self.parent.index(-1)

    # ------------------------------------------------------------- traversal
    def traversal(self) -> list[int]:
        """Link indices in an order where every parent precedes its children (stable DFS
        in document order of the children)."""
        L = self.num_links
        children: list[list[int]] = [[] for _ in range(L)]
        for l, p in enumerate(self.parent):
            if p >= 0:
                children[p].append(l)
        order: list[int] = []
        stack = [self.base_index]
        while stack:
            l = stack.pop()
            order.append(l)
            stack.extend(reversed(children[l]))
        if len(order) != L:
            raise ValueError("kinematic structure is not a single tree")
        return order

    def ancestors_dofs(self) -> list[list[int]]:
        """For each link the DOF indices of the movable joints on its path to the base
        (root side first)."""
        out: list[list[int]] = [[] for _ in range(self.num_links)]
        for l in self.traversal():
            p = self.parent[l]
            if p >= 0:
                out[l] = list(out[p])
                if self.dof_index[l] >= 0:
                    out[l].append(self.dof_index[l])
        return out

    def x_std(self) -> np.ndarray:
        """Flat a-priori standard parameter vector (10 per link, link-major)."""
        return np.ascontiguousarray(self.params, dtype=np.float64).reshape(-1).copy()

    # ------------------------------------------------------------- serialisation
    def _ranks(self) -> tuple[list[int], list[int]]:
        L = self.num_links
        lr = list(self.doc_link_rank) if len(self.doc_link_rank) == L else list(range(L))
        jr = list(self.doc_joint_rank) if len(self.doc_joint_rank) == L else [-1 if p < 0 else l for l, p in enumerate(self.parent)]
        return lr, jr

    def visit_order(self) -> list[int]:
        """Link indices in iDynTree's visiting order: depth-first from the base, the children of a link pushed on a
        LIFO stack in the document order of their joints, a link numbered when popped (so the child whose joint
        comes LAST in the URDF is walked first)."""
        _, jr = self._ranks()
        children: list[list[int]] = [[] for _ in range(self.num_links)]
        for l, p in enumerate(self.parent):
            if p >= 0:
                children[p].append(l)
        order: list[int] = []
        stack = [self.base_index]
        while stack:
            l = stack.pop()
            order.append(l)
            stack.extend(sorted(children[l], key=lambda c: jr[c]))
        if len(order) != self.num_links:
            raise ValueError("kinematic structure is not a single tree")
        return order

    def link_order(self, mode: str) -> list[int]:
        """Current link indices in the order of serialisation ``mode`` ("traversal" | "document")."""
        if mode == "traversal":
            return self.visit_order()
        if mode == "document":
            lr, _ = self._ranks()
            return sorted(range(self.num_links), key=lambda l: lr[l])
        raise ValueError(f"unknown link order '{mode}' (traversal | document)")

    def dof_order(self, mode: str) -> list[str]:
        """Names of the movable joints in the order of serialisation ``mode``."""
        if mode == "traversal":
            return [self.joint_names[l] for l in self.visit_order() if self.dof_index[l] >= 0]
        if mode == "document":
            _, jr = self._ranks()
            return [self.joint_names[l] for l in sorted(range(self.num_links), key=lambda l: jr[l]) if self.dof_index[l] >= 0]
        raise ValueError(f"unknown DOF order '{mode}' (traversal | document)")

    def reordered_links(self, link_names: list[str]) -> "Topology":
        """Same tree with the links (= the 10-column parameter blocks) serialised as ``link_names``."""
        if sorted(link_names) != sorted(self.link_names) or len(set(link_names)) != len(link_names):
            raise ValueError("link_names must be a permutation of the model's links")
        old = [self.link_names.index(n) for n in link_names]  # new index -> old index
        new_of = {o: i for i, o in enumerate(old)}
        lr, jr = self._ranks()
        return Topology(
            name=self.name,
            link_names=list(link_names),
            parent=[new_of[self.parent[o]] if self.parent[o] >= 0 else -1 for o in old],
            joint_names=[self.joint_names[o] for o in old],
            joint_type=[self.joint_type[o] for o in old],
            dof_index=[self.dof_index[o] for o in old],
            rest_R=np.asarray(self.rest_R)[old].copy(),
            rest_p=np.asarray(self.rest_p)[old].copy(),
            axis=np.asarray(self.axis)[old].copy(),
            params=np.asarray(self.params)[old].copy(),
            dof_names=list(self.dof_names),
            limits={k: dict(v) for k, v in self.limits.items()},
            friction={k: dict(v) for k, v in self.friction.items()},
            frames={k: {"link": new_of[v["link"]], "R": np.array(v["R"], dtype=float), "p": np.array(v["p"], dtype=float)}
                    for k, v in self.frames.items()},
            doc_link_rank=[lr[o] for o in old],
            doc_joint_rank=[jr[o] for o in old],
        )

    def serialized(self, link_order: str | None = "traversal", dof_order: str | None = "traversal",
                   joint_names: list[str] | None = None, link_names: list[str] | None = None) -> "Topology":
        """The same robot in the requested serialisation (``None`` keeps what this object has); explicit name lists win."""
        t = self
        names = link_names if link_names is not None else (None if link_order is None else [t.link_names[l] for l in t.link_order(link_order)])
        if names is not None and list(names) != t.link_names:
            t = t.reordered_links(list(names))
        dofs = joint_names if joint_names is not None else (None if dof_order is None else t.dof_order(dof_order))
        if dofs is not None and list(dofs) != t.dof_names:
            t = t.reordered_dofs(list(dofs))
        return t

    def reordered_dofs(self, joint_names: list[str]) -> "Topology":
        """Same tree with the DOF serialisation given by ``joint_names``
        (the reference takes it from the regressor XML, ``model.py:74-85``)."""
        if sorted(joint_names) != sorted(self.dof_names):
            raise ValueError("joint_names must be a permutation of the model's movable joints")
        new_index = {n: i for i, n in enumerate(joint_names)}
        t = Topology.from_dict(self.to_dict())
        t.dof_names = list(joint_names)
        t.dof_index = [new_index[self.joint_names[l]] if d >= 0 else -1 for l, d in enumerate(self.dof_index)]
        return t

    # ---------------------------------------------------------------- (de)serialise
    def to_dict(self) -> dict[str, Any]:
        return {
            "name": self.name,
            "link_names": self.link_names,
            "parent": self.parent,
            "joint_names": self.joint_names,
            "joint_type": self.joint_type,
            "dof_index": self.dof_index,
            "rest_R": np.asarray(self.rest_R).tolist(),
            "rest_p": np.asarray(self.rest_p).tolist(),
            "axis": np.asarray(self.axis).tolist(),
            "params": np.asarray(self.params).tolist(),
            "dof_names": self.dof_names,
            "limits": self.limits,
            "friction": self.friction,
            "frames": {
                k: {"link": v["link"], "R": np.asarray(v["R"]).tolist(), "p": np.asarray(v["p"]).tolist()}
                for k, v in self.frames.items()
            },
            "doc_link_rank": list(self._ranks()[0]),
            "doc_joint_rank": list(self._ranks()[1]),
        }

    @staticmethod
    def from_dict(d: dict[str, Any]) -> "Topology":
        return Topology(
            name=d["name"],
            link_names=list(d["link_names"]),
            parent=[int(v) for v in d["parent"]],
            joint_names=list(d["joint_names"]),
            joint_type=[int(v) for v in d["joint_type"]],
            dof_index=[int(v) for v in d["dof_index"]],
            rest_R=np.array(d["rest_R"], dtype=np.float64).reshape(-1, 3, 3),
            rest_p=np.array(d["rest_p"], dtype=np.float64).reshape(-1, 3),
            axis=np.array(d["axis"], dtype=np.float64).reshape(-1, 3),
            params=np.array(d["params"], dtype=np.float64).reshape(-1, 10),
            dof_names=list(d["dof_names"]),
            limits={k: dict(v) for k, v in d.get("limits", {}).items()},
            friction={k: dict(v) for k, v in d.get("friction", {}).items()},
            frames={
                k: {"link": int(v["link"]), "R": np.array(v["R"], dtype=float), "p": np.array(v["p"], dtype=float)}
                for k, v in d.get("frames", {}).items()
            },
            doc_link_rank=[int(v) for v in d.get("doc_link_rank", [])],
            doc_joint_rank=[int(v) for v in d.get("doc_joint_rank", [])],
        )

    def save_json(self, path: str) -> None:
        with open(path, "w") as f:
            json.dump(self.to_dict(), f)

    @staticmethod
    def load_json(path: str) -> "Topology":
        with open(path) as f:
            return Topology.from_dict(json.load(f))

    @staticmethod
    def load(path: str, link_order: str | None = None, dof_order: str | None = None) -> "Topology":
        """Load from a URDF (``.urdf``/xml; default serialisation "traversal") or from a topology JSON written by
        ``save_json`` (keeps the stored serialisation unless an order is asked for)."""
        if path.endswith(".json"):
            t = Topology.load_json(path)
            return t.serialized(link_order, dof_order) if (link_order or dof_order) else t
        return parse_urdf(path, link_order=link_order or "traversal", dof_order=dof_order or "traversal")


def parse_urdf(path: str, joint_names: list[str] | None = None, link_order: str = "traversal",
               dof_order: str = "traversal", link_names: list[str] | None = None) -> Topology:
    """Extract the topology from a URDF file (rules in the module docstring)."""
    root = ET.parse(path).getroot()

    # ---- raw links (document order) ----
    raw_links: list[str] = []
    raw_inertial: dict[str, tuple[float, np.ndarray, np.ndarray]] = {}
    for le in root.findall("link"):
        name = le.attrib["name"]
        raw_links.append(name)
        ie = le.find("inertial")
        mass = 0.0
        com = np.zeros(3)
        I_c = np.zeros((3, 3))
        if ie is not None:
            me = ie.find("mass")
            if me is not None:
                mass = float(me.attrib.get("value", 0.0))
            R_i, com = _origin(ie)
            ine = ie.find("inertia")
            if ine is not None:
                a = {k: float(ine.attrib.get(k, 0.0)) for k in ("ixx", "ixy", "ixz", "iyy", "iyz", "izz")}
                I_local = np.array(
                    [[a["ixx"], a["ixy"], a["ixz"]], [a["ixy"], a["iyy"], a["iyz"]], [a["ixz"], a["iyz"], a["izz"]]]
                )
                I_c = R_i @ I_local @ R_i.T
        raw_inertial[name] = (mass, com, I_c)

    # ---- raw joints (document order) ----
    raw_joints = []
    for je in root.findall("joint"):
        jtype = je.attrib.get("type", "fixed")
        if jtype in ("revolute", "continuous"):
            t = JOINT_REVOLUTE
        elif jtype == "prismatic":
            t = JOINT_PRISMATIC
        elif jtype == "fixed":
            t = JOINT_FIXED
        else:  # floating / planar joints: several DOFs per joint, which neither the reference's per-joint friction layout nor this path has
            raise NotImplementedError(f"joint type '{jtype}' of joint {je.attrib.get('name')} is not supported")
        R, p = _origin(je)
        ax_e = je.find("axis")
        ax = _vec(ax_e.attrib.get("xyz") if ax_e is not None else None, [1, 0, 0])
        if t != JOINT_FIXED:
            nrm = np.linalg.norm(ax)
            if nrm == 0:
                raise ValueError(f"joint {je.attrib['name']} has a zero axis")
            ax = ax / nrm
        else:
            ax = np.zeros(3)
        raw_joints.append(
            {
                "name": je.attrib["name"],
                "type": t,
                "parent": je.find("parent").attrib["link"],
                "child": je.find("child").attrib["link"],
                "R": R,
                "p": p,
                "axis": ax,
                "elem": je,
            }
        )

    # ---- neighbours, fake links (decided on the ORIGINAL graph, not iteratively) ----
    neigh: dict[str, list[dict]] = {n: [] for n in raw_links}
    for j in raw_joints:
        neigh[j["parent"]].append(j)
        neigh[j["child"]].append(j)
    child_names = {j["child"] for j in raw_joints}
    roots = [n for n in raw_links if n not in child_names]
    if len(roots) != 1:
        raise ValueError(f"URDF must have exactly one root link, found {roots}")

    def is_fake(n: str) -> bool:
        return raw_inertial[n][0] == 0.0 and len(neigh[n]) == 1 and neigh[n][0]["type"] == JOINT_FIXED

    fake = {n for n in raw_links if is_fake(n)}
    kept = [n for n in raw_links if n not in fake]
    if not kept:
        raise ValueError("no links with inertia in URDF")
    idx = {n: i for i, n in enumerate(kept)}
    base = roots[0]
    if base in fake:
        base = neigh[base][0]["child"]
        if base in fake:
            raise ValueError("fake root link attached to another fake link")

    L = len(kept)
    parent = [-2] * L
    joint_name = [""] * L
    joint_type = [JOINT_FIXED] * L
    rest_R = np.tile(np.eye(3), (L, 1, 1))
    rest_p = np.zeros((L, 3))
    axis = np.zeros((L, 3))
    frames: dict[str, dict[str, Any]] = {}
    movable: list[str] = []
    for j in raw_joints:
        if j["child"] in fake:
            # leaf fake link -> frame on its parent
            if j["parent"] in idx:
                frames[j["child"]] = {"link": idx[j["parent"]], "R": j["R"], "p": j["p"]}
            continue
        if j["parent"] in fake:
            # fake root: its frame expressed in the new base (inverse of the joint origin)
            frames[j["parent"]] = {"link": idx[j["child"]], "R": j["R"].T, "p": -j["R"].T @ j["p"]}
            continue
        c = idx[j["child"]]
        parent[c] = idx[j["parent"]]
        joint_name[c] = j["name"]
        joint_type[c] = j["type"]
        rest_R[c] = j["R"]
        rest_p[c] = j["p"]
        axis[c] = j["axis"]
        if j["type"] != JOINT_FIXED:
            movable.append(j["name"])
    parent[idx[base]] = -1
    if any(p == -2 for p in parent):
        raise ValueError("disconnected links in URDF")

    if joint_names is not None and sorted(joint_names) != sorted(movable):
        raise ValueError("joint_names must list exactly the movable joints of the URDF")
    dof_names = list(movable)  # document order here; serialised below
    dof_of = {n: i for i, n in enumerate(dof_names)}
    dof_index = [dof_of[joint_name[l]] if joint_type[l] != JOINT_FIXED else -1 for l in range(L)]

    params = np.stack([inertial_to_params(*raw_inertial[n]) for n in kept])

    # limits / friction exactly as helpers.getJointLimits / getJointFriction read them
    limits: dict[str, dict[str, float]] = {}
    friction: dict[str, dict[str, float]] = {}
    for j in raw_joints:
        je = j["elem"]
        # (the reference reads limits / friction of revolute joints only, helpers.py:907,957; prismatic joints are read the same way here
        # so that the random-state generator of getRandomRegressor has their ranges)
        if je.attrib.get("type") not in ("revolute", "prismatic"):
            continue
        le = je.find("limit")
        if le is not None:
            limits[j["name"]] = {
                "torque": float(le.attrib["effort"]),
                "lower": float(le.attrib["lower"]),
                "upper": float(le.attrib["upper"]),
                "velocity": float(le.attrib["velocity"]),
            }
        de = je.find("dynamics")
        fc = fv = 0.0
        if de is not None:
            fc = float(de.attrib.get("friction", 0.0))
            fv = float(de.attrib.get("damping", 0.0))
        friction[j["name"]] = {"f_constant": fc, "f_velocity": fv}

    jrank = {j["name"]: r for r, j in enumerate(raw_joints)}
    doc = Topology(
        name=root.attrib.get("name", ""),
        link_names=kept,
        parent=parent,
        joint_names=joint_name,
        joint_type=joint_type,
        dof_index=dof_index,
        rest_R=rest_R,
        rest_p=rest_p,
        axis=axis,
        params=params,
        dof_names=dof_names,
        limits=limits,
        friction=friction,
        frames=frames,
        doc_link_rank=list(range(L)),
        doc_joint_rank=[jrank[joint_name[l]] if parent[l] >= 0 else -1 for l in range(L)],
    )
    return doc.serialized(link_order, dof_order, joint_names=joint_names, link_names=link_names)


# ---------------------------------------------------------------------------------------------------------
# On-disk side after the path: identified standard parameters back into a URDF
# ---------------------------------------------------------------------------------------------------------
def params_link_to_bary(params: np.ndarray, num_links: int) -> np.ndarray:
    """``ParamHelpers.paramsLink2Bary`` (identification/helpers.py:374-407): per link [m, m c, I about the link origin]
    -> [m, c, I about the COM] (what a URDF ``<inertial>`` holds); trailing friction entries are left alone."""
    out = np.array(params, dtype=float, copy=True)
    for l in range(num_links):
        p = out[10 * l:10 * l + 10]
        m = p[0]
        c = p[1:4] / m if m != 0 else np.zeros(3)
        I_o = np.array([[p[4], p[5], p[6]], [p[5], p[7], p[8]], [p[6], p[8], p[9]]])
        I_c = I_o - m * (float(c @ c) * np.eye(3) - np.outer(c, c))
        p[1:4] = c
        p[4:10] = [I_c[0, 0], I_c[0, 1], I_c[0, 2], I_c[1, 1], I_c[1, 2], I_c[2, 2]]
    return out


def params_bary_to_link(params: np.ndarray, num_links: int) -> np.ndarray:
    """``ParamHelpers.paramsBary2Link`` (identification/helpers.py:409-435), the inverse of ``params_link_to_bary``."""
    out = np.array(params, dtype=float, copy=True)
    for l in range(num_links):
        p = out[10 * l:10 * l + 10]
        I_c = np.array([[p[4], p[5], p[6]], [p[5], p[7], p[8]], [p[6], p[8], p[9]]])
        out[10 * l:10 * l + 10] = inertial_to_params(p[0], p[1:4].copy(), I_c)
    return out


def replace_params_in_urdf(input_urdf: str, output_urdf: str, topo: "Topology", new_params: np.ndarray,
                           friction_layout: dict[str, Any] | None = None) -> None:
    """``URDFHelpers.replaceParamsInURDF`` (identification/helpers.py:511-577): write the standard parameters
    ``new_params`` (link-frame convention, 10 per kept link, then friction slots) into a copy of ``input_urdf``:
    mass, COM (``origin xyz``), inertia about the COM, and per joint ``dynamics friction`` (Coulomb) / ``damping``
    (symmetric viscous).  ``friction_layout`` = {"coulomb_offset": index of F_c of joint 0, "viscous_offset": index of
    F_v of joint 0 or None}; without it the joint dynamics are written as 0 like the reference does when friction was
    not identified.  Like the reference, the ``<inertial><origin rpy>`` of the input is kept: inputs whose inertial frame
    is rotated get ``rpy`` reset to zero here (the written tensor is in link axes), which the reference silently skips."""
    import xml.etree.ElementTree as ET

    x = params_link_to_bary(np.asarray(new_params, dtype=float), topo.num_links)
    tree = ET.parse(input_urdf)
    root = tree.getroot()
    for le in root.findall("link"):
        name = le.attrib.get("name")
        if name not in topo.link_names:
            continue
        i = topo.link_names.index(name)
        b = x[10 * i:10 * i + 10]
        me = le.find("inertial/mass")
        if me is not None:
            me.attrib["value"] = repr(float(b[0]))
        oe = le.find("inertial/origin")
        if oe is not None:
            oe.attrib["xyz"] = f"{float(b[1])!r} {float(b[2])!r} {float(b[3])!r}"
            if "rpy" in oe.attrib:
                oe.attrib["rpy"] = "0 0 0"
        ie = le.find("inertial/inertia")
        if ie is not None:
            for key, v in zip(("ixx", "ixy", "ixz", "iyy", "iyz", "izz"), b[4:10]):
                ie.attrib[key] = repr(float(v))
    for je in root.findall("joint"):
        name = je.attrib.get("name")
        if name not in topo.dof_names:
            continue
        j = topo.dof_names.index(name)
        fc = fv = 0.0
        if friction_layout is not None:
            fc = float(x[friction_layout["coulomb_offset"] + j])
            if friction_layout.get("viscous_offset") is not None:
                fv = float(x[friction_layout["viscous_offset"] + j])
        de = je.find("dynamics")
        if de is not None:
            de.attrib["friction"] = repr(fc)
            de.attrib["damping"] = repr(fv)
    tree.write(output_urdf, xml_declaration=True)
