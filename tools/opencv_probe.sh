#!/bin/bash
# Is there a real OpenCV anywhere on the GPU box (or reachable from it)?  VERDICT r04 item 1: the claim "the GPU boxes
# have no cv2" had never been checked.  Usage:
#   gpurun --timeout 600 -- 'bash tools/opencv_probe.sh'
# Writes gpurun_out/opencv_probe/probe.txt (copied to profiles/r05_opencv_probe.txt).  If any route yields cv2, runs
# tools/opencv_crosscheck.py --write-golden and copies tests/golden/opencv_*.npz + the build information back.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/opencv_probe
mkdir -p "$OUT"
cd "$ROOT" || exit 1
export TMPDIR=/tmp
{
echo "== date / host"; date -u; uname -a; head -2 /etc/os-release
echo "== python"; which -a python python3; python --version
echo "== import cv2 (every interpreter on PATH and under /opt, /usr)"
for PY in $(ls /usr/bin/python3* /usr/local/bin/python3* /opt/*/bin/python3* /opt/conda/bin/python* 2>/dev/null | sort -u); do
    echo "-- $PY"; timeout 60 "$PY" -c "import cv2; print(cv2.__version__); print(cv2.getBuildInformation())" 2>&1 | head -100
done
echo "== pkg-config"; pkg-config --modversion opencv4 2>&1; pkg-config --modversion opencv 2>&1
echo "== find libopencv* / cv2*.so / opencv headers / wheels"
timeout 240 find / -xdev \( -name 'libopencv*' -o -name 'cv2*.so' -o -name 'cv2' -o -name 'opencv2' -o -name 'opencv*.whl' -o -name 'opencv*.tar*' -o -name 'opencv_python*' -o -name 'OpenCVConfig*.cmake' \) -not -path '/proc/*' -not -path "$ROOT/*" 2>/dev/null | head -50
echo "== other mounts"; mount | grep -v -E 'proc|sysfs|cgroup|devpts|tmpfs|mqueue' | head -20
for M in $(mount | awk '{print $3}' | grep -v -E '^/(proc|sys|dev)' | sort -u); do
    [ "$M" = "/" ] && continue
    timeout 60 find "$M" -xdev \( -name 'libopencv*' -o -name 'cv2*.so' -o -name 'opencv*.whl' \) 2>/dev/null | head -5
done
echo "== conda / pip / dpkg listings"
(conda list 2>/dev/null | grep -i opencv) || echo "(no conda or no opencv in it)"
(python -m pip list 2>/dev/null | grep -i -E 'opencv|cv2|kornia|scikit-image|imageio|pillow|torchvision|mmcv') || echo "(pip list: nothing image-related)"
(dpkg -l 2>/dev/null | grep -i -E 'opencv|libcv') || echo "(dpkg: no opencv)"
(apt-cache policy libopencv-dev python3-opencv 2>/dev/null | head -12) || true
echo "== libraries that embed OpenCV code (MIOpen / MIVisionX / rocAL / rpp ship cv-like kernels, not OpenCV itself)"
ls /opt/rocm/lib 2>/dev/null | grep -i -E 'opencv|vx_|rocal|rpp|mivision' | head
echo "== torchvision ops that could cross-check (none implement calcOpticalFlowPyrLK)"
python -c "import torchvision; print('torchvision', torchvision.__version__)" 2>&1 | tail -1
python -c "import skimage; print('skimage', skimage.__version__)" 2>&1 | tail -1
python -c "import kornia; print('kornia', kornia.__version__)" 2>&1 | tail -1
echo "== network: can the lease reach a package index?"
cat /etc/resolv.conf 2>/dev/null | head -5
env | grep -i -E 'proxy|pip_|index' || echo "(no proxy / pip env)"
cat /etc/pip.conf ~/.pip/pip.conf ~/.config/pip/pip.conf 2>/dev/null
timeout 20 getent hosts pypi.org files.pythonhosted.org 2>&1 || echo "(pypi.org does not resolve)"
timeout 30 python - <<'PYEOF'
import socket
for host, port in (("pypi.org", 443), ("files.pythonhosted.org", 443), ("151.101.0.223", 443), ("1.1.1.1", 443), ("github.com", 443)):
    try:
        s = socket.create_connection((host, port), timeout=5); s.close(); print("connect %s:%d ok" % (host, port))
    except Exception as e:
        print("connect %s:%d failed: %r" % (host, port, e))
PYEOF
mkdir -p /tmp/w /tmp/cv
echo "-- pip download opencv-python-headless==4.5.5.64"
timeout 120 python -m pip download --no-deps --disable-pip-version-check --timeout 10 --retries 1 opencv-python-headless==4.5.5.64 -d /tmp/w 2>&1 | tail -6
echo "-- pip download opencv-python-headless (any)"
timeout 120 python -m pip download --no-deps --disable-pip-version-check --timeout 10 --retries 1 opencv-python-headless -d /tmp/w 2>&1 | tail -6
ls -la /tmp/w
echo "-- wheelhouse / find-links directories on the box"
timeout 60 find / -xdev -type d \( -name 'wheelhouse' -o -name 'wheels' \) -not -path '/proc/*' 2>/dev/null | head
} > "$OUT/probe.txt" 2>&1

CVPATH=""
if python -c "import cv2" 2>/dev/null; then
    CVPATH="(system)"
elif ls /tmp/w/*.whl >/dev/null 2>&1; then
    python -m pip install --no-deps --no-index --target /tmp/cv /tmp/w/*.whl >> "$OUT/probe.txt" 2>&1 && CVPATH=/tmp/cv
fi
if [ -n "$CVPATH" ]; then
    echo "== a real OpenCV is available via $CVPATH: running tools/opencv_crosscheck.py --write-golden" >> "$OUT/probe.txt"
    [ "$CVPATH" != "(system)" ] && export PYTHONPATH=/tmp/cv:$PYTHONPATH
    python -c "import cv2; print(cv2.__version__); print(cv2.getBuildInformation())" > "$OUT/opencv_build_information.txt" 2>&1
    timeout 900 python tools/opencv_crosscheck.py --write-golden > "$OUT/crosscheck.txt" 2>&1
    echo "crosscheck rc=$?" >> "$OUT/probe.txt"
    cp tests/golden/opencv_*.npz "$OUT/" 2>/dev/null
    tail -60 "$OUT/crosscheck.txt"
else
    echo "== RESULT: no OpenCV on the GPU box by any route (import, pkg-config, filesystem, package index)" >> "$OUT/probe.txt"
fi
tail -70 "$OUT/probe.txt"
