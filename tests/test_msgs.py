"""The ROS wire schemas carried under msg/ must stay byte-identical to the reference's (msg/*.msg): a ROS message type
is identified by the MD5 of its definition text.  The fixture hashes below were taken from /root/reference/msg in the
authoring container (md5sum); when the reference tree is present the files are also compared byte for byte."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MD5 = {
    "awareness.msg": "24d5935c6556fff6b58b8c94dd4db057",
    "awareness2local.msg": "c935299ceca2b20a12122c476ca2fa2b",
    "esdfs.msg": "877bf0a424eff3d536bb888f37bec067",
    "localmap.msg": "47e9927b99d93c4d24df3062adec41ef",
}


def test_msg_files_are_the_reference_schemas():
    for name, want in MD5.items():
        data = open(os.path.join(ROOT, "msg", name), "rb").read()
        assert hashlib.md5(data).hexdigest() == want, name
        ref = os.path.join("/root/reference/msg", name)
        if os.path.exists(ref):  # authoring container only; the GPU box has no reference tree
            assert open(ref, "rb").read() == data, name
    assert sorted(f for f in os.listdir(os.path.join(ROOT, "msg")) if f.endswith(".msg")) == sorted(MD5)


def test_msg_fields_cover_what_the_boundary_exports():
    """awareness.msg / localmap.msg carry (T, count, uint32[] idx): the counts and index lists the C ABI exposes through
    mlm_frame_stats / mlm_get_awareness_hits — same field names and types as the reference."""
    aw = open(os.path.join(ROOT, "msg", "awareness.msg")).read().split()
    assert aw == ["Header", "header", "geometry_msgs/Transform", "T_w_a", "uint32", "occupied_cell_count", "uint32[]",
                  "occupied_cell_idx"]
    lm = open(os.path.join(ROOT, "msg", "localmap.msg")).read().split()
    assert lm == ["Header", "header", "geometry_msgs/Transform", "T_w_l", "uint32", "occupied_cell_count", "uint32[]",
                  "occupied_cell_idx"]
