"""Numbers of the reference's quadruped asset (newton/examples/assets/quadruped.urdf), read with ElementTree -- independent of
newton_amd.urdf -- into tests/golden/quadruped_asset.json.  tests/test_host_logic.py::test_quadruped_scene_is_the_reference_asset
parses tests/scenes.py's generated URDF the same way and compares: the bench / test scene IS the reference geometry.
Run from the repo root in the build container (needs /root/reference):  python tests/golden/make_quadruped_asset_vectors.py"""
import json
import os
import xml.etree.ElementTree as ET

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/newton/examples/assets/quadruped.urdf"


def _floats(s):
    return [float(x) for x in s.split()]


def asset_numbers(xml_text: str) -> dict:
    root = ET.fromstring(xml_text)
    joints, links = [], []
    for j in root.findall("joint"):
        o = j.find("origin")
        lim = j.find("limit")
        joints.append(dict(name=j.get("name"), type=j.get("type"), parent=j.find("parent").get("link"),
                           child=j.find("child").get("link"), rpy=_floats(o.get("rpy")), xyz=_floats(o.get("xyz")),
                           axis=_floats(j.find("axis").get("xyz")), effort=float(lim.get("effort")),
                           velocity=float(lim.get("velocity"))))
    for link in root.findall("link"):
        c = link.find("collision")
        o = c.find("origin")
        g = list(c.find("geometry"))[0]
        links.append(dict(name=link.get("name"), rpy=_floats(o.get("rpy")), xyz=_floats(o.get("xyz")), geometry=g.tag,
                          dims={k: float(v) for k, v in g.attrib.items()}))
    return dict(joints=joints, links=links)


if __name__ == "__main__":
    with open(SRC, encoding="utf-8") as f:
        numbers = asset_numbers(f.read().encode("utf-8").decode("utf-8"))
    with open(os.path.join(HERE, "quadruped_asset.json"), "w") as f:
        json.dump(numbers, f, indent=1, sort_keys=True)
    print("wrote", len(numbers["joints"]), "joints,", len(numbers["links"]), "links")
