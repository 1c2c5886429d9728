"""CPU restatement of SfM::saveCloudAndCamerasToPLY (SfMToyLib/SfM.cpp:631-711) -- TEST INFRASTRUCTURE ONLY.

Byte-exact: the reference writes through std::ofstream with default formatting, i.e. every float / double goes through
printf's %g with precision 6 (Python's '%g' is the same conversion), integers as decimal, std::endl = one newline.  The
header literals carry the reference's trailing blanks (SfM.cpp:639-648, 670-682); every vertex record of the point file
ends with a blank before the newline (SfM.cpp:659-664).  Colours: image[row][col] of the FIRST originating view's feature,
the float coordinates converted with cv::saturate_cast<int> == round half to even (SfM.cpp:652-656), written R G B from a
BGR pixel (SfM.cpp:662-664).

parity pinned?  No reference test covers the function and OpenCV is not available here: "parity unpinned"; the format
rules above are the C++ standard's and OpenCV's documented conversions, checked here against a hand-written expected file
(tests/test_ply_export.py)."""
import numpy as np

POINT_HEADER = ["ply                 ", "format ascii 1.0    ", "element vertex %d", "property float x    ", "property float y    ",
                "property float z    ", "property uchar red  ", "property uchar green", "property uchar blue ", "end_header          "]
CAMERA_HEADER = ["ply                 ", "format ascii 1.0    ", "element vertex %d", "property float x    ", "property float y    ",
                 "property float z    ", "element edge %d", "property int vertex1", "property int vertex2", "property uchar red  ",
                 "property uchar green", "property uchar blue ", "end_header          "]


def g(v):
    return "%g" % float(v)


def points_ply(cloud, feats, images):
    """cloud: list of (xyz float32[3], {view: feature}); feats: list of [n,2] float32; images: list of [rows, cols, 3] uint8 BGR."""
    out = []
    for l in POINT_HEADER:
        out.append(l % len(cloud) if "%d" in l else l)
    for xyz, views in cloud:
        v = min(views)                                           # std::map::begin()
        x, y = feats[v][views[v]]
        col, row = int(np.rint(np.float64(np.float32(x)))), int(np.rint(np.float64(np.float32(y))))      # cvRound: half to even
        b, gr, r = (int(c) for c in images[v][row, col])
        p = np.asarray(xyz, np.float32)
        out.append("%s %s %s %d %d %d " % (g(p[0]), g(p[1]), g(p[2]), r, gr, b))
    return "\n".join(out) + "\n"


def cameras_ply(poses):
    """poses: [n, 3, 4] float32."""
    poses = np.asarray(poses, np.float32)
    n = len(poses)
    out = []
    for l in CAMERA_HEADER:
        out.append(l % (4 * n if "vertex" in l else 3 * n) if "%d" in l else l)
    for P in poses:
        c = P[:, 3].astype(np.float64)
        out.append("%s %s %s" % (g(c[0]), g(c[1]), g(c[2])))
        for axis in range(3):
            tip = c + P[:, axis].astype(np.float64) * 0.2        # Point3d arithmetic (SfM.cpp:685-688)
            out.append("%s %s %s" % (g(tip[0]), g(tip[1]), g(tip[2])))
    for i in range(n):
        for axis, colour in enumerate(("255 0 0", "0 255 0", "0 0 255")):
            out.append("%d %d %s" % (4 * i, 4 * i + 1 + axis, colour))
    return "\n".join(out) + "\n"
