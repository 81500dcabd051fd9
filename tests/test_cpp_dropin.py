"""The C++ drop-in (lsqrrecipes_amd/include/*.h, same class / method names as the reference's
headers) and the example programs: they must compile and link on CPU; on the GPU the programs run
and check themselves (tests/cpp/estimatorTests.cxx restates the reference's ctest assertions)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "examples", "build")
PROGS = ["planeEstimation", "sphereEstimation", "lineEstimation", "linearEquationSystemSolver",
         "crosswireUSCalibration", "AbsoluteOrientation", "pivotCalibration",
         "rayIntersectionEstimation", "planeUSCalibration", "estimatorTests"]
REFDATA = os.path.join(ROOT, "tests", "golden", "ref_data")


def test_dropin_headers_compile_and_link():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    for p in PROGS:
        assert os.access(os.path.join(BUILD, p), os.X_OK)


def test_reference_header_names_present():
    """a source written against the reference includes these names"""
    inc = os.path.join(ROOT, "lsqrrecipes_amd", "include")
    for h in ["RANSAC.h", "ParametersEstimator.h", "PlaneParametersEstimator.h",
              "SphereParametersEstimator.h", "LineParametersEstimator.h",
              "DenseLinearEquationSystemParametersEstimator.h",
              "SinglePointTargetUSCalibrationParametersEstimator.h",
              "PlanePhantomUSCalibrationParametersEstimator.h",
              "AbsoluteOrientationParametersEstimator.h", "PivotCalibrationParametersEstimator.h",
              "RayIntersectionParametersEstimator.h", "Ray3D.h", "Vector3D.h",
              "Line2DParametersEstimator.h",
              "Point.h", "Point2D.h",
              "Point3D.h", "Frame.h", "Epsilon.h", "copyright.h"]:
        assert os.path.exists(os.path.join(inc, h)), h


def _run(args):
    if not os.path.exists(os.path.join(BUILD, args[0])):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")],
                              stdout=subprocess.DEVNULL)
    env = dict(os.environ)
    env.setdefault("LSQR_OIV_DIR", os.path.join(BUILD, "scenes"))  # .iv scenes of the 3-D examples
    os.makedirs(env["LSQR_OIV_DIR"], exist_ok=True)
    r = subprocess.run([os.path.join(BUILD, args[0])] + args[1:], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    return r.stdout


@pytest.mark.gpu
def test_reference_style_estimator_tests_on_gpu():
    out = _run(["estimatorTests", os.path.join(REFDATA, "augmentedMatrix.txt"),
                os.path.join(REFDATA, "pivotCalibrationData.txt")])
    assert "all checks passed" in out


@pytest.mark.gpu
@pytest.mark.parametrize("prog", ["planeEstimation", "sphereEstimation", "lineEstimation",
                                  "pivotCalibration", "rayIntersectionEstimation",
                                  "planeUSCalibration"])
def test_example_programs_on_gpu(prog):
    out = _run([prog])
    assert "RANSAC" in out


@pytest.mark.gpu
def test_section_8f_example_programs_on_gpu():
    out = _run(["AbsoluteOrientation"])
    assert "Exhaustive search transformation" in out
    out = _run(["pivotCalibration", os.path.join(REFDATA, "pivotCalibrationDataWithOutliers.txt")])
    assert "RANSAC translations" in out


@pytest.mark.gpu
def test_example_data_file_programs_on_gpu():
    out = _run(["linearEquationSystemSolver", os.path.join(REFDATA, "augmentedMatrixWithOutliers.txt")])
    assert "Experimental data, RANSAC solution" in out
    out = _run(["crosswireUSCalibration"])
    assert "Percentage of frames used" in out
    r = subprocess.run([os.path.join(BUILD, "crosswireUSCalibration"),
                        os.path.join(REFDATA, "crossWirePhantomTransformations.txt"),
                        os.path.join(REFDATA, "crossWirePhantom2DPoints.txt")],
                       capture_output=True, text=True, timeout=300)
    assert "54 frames" in r.stdout


@pytest.mark.gpu
def test_crosswire_writes_igstk_xml(tmp_path):
    """the reference example's output wire format (examples/crosswireUSCalibration.cxx:181-210)"""
    xml = tmp_path / "calibration.xml"
    r = subprocess.run([os.path.join(BUILD, "crosswireUSCalibration"),
                        os.path.join(REFDATA, "crossWirePhantomTransformations.txt"),
                        os.path.join(REFDATA, "crossWirePhantom2DPoints.txt"), str(xml)],
                       capture_output=True, text=True, timeout=300)
    assert "54 frames" in r.stdout
    if r.returncode == 0:
        import xml.etree.ElementTree as ET
        root = ET.parse(str(xml)).getroot()
        assert root.tag == "precomputed_transform"
        tr = root.find("transformation")
        assert float(tr.attrib["estimation_error"]) >= 0
        assert len(tr.text.split()) == 12


@pytest.mark.gpu
@pytest.mark.parametrize("prog,shape", [("planeEstimation", "IndexedFaceSet"), ("sphereEstimation", "Sphere"),
                                        ("lineEstimation", "LineSet")])
def test_examples_write_open_inventor_scenes(prog, shape):
    """the reference's 3-D examples save their fits as Open Inventor scenes
    (examples/planeEstimation.cxx:205-330): one ball per observation, green inside the consensus set,
    plus the estimated model"""
    _run([prog])
    stem = prog.replace("Estimation", "").capitalize()
    for kind in ("leastSquares", "RANSAC"):
        path = os.path.join(BUILD, "scenes", "%s%sEstimation.iv" % (kind, stem))
        text = open(path).read()
        assert text.startswith("#Inventor V2.1 ascii")
        assert text.count("Separator {") == 101 and text.count("Transform {") >= 100   # 100 observations + the model
        assert text.count("{") == text.count("}")
        assert shape in text.split("Separator {")[-1]
        if kind == "RANSAC":   # consensus set green (noise 0.4 against delta 0.5: most of the 90), outliers red
            green, red = text.count("ambientColor 0.0 1.0 0.0"), text.count("ambientColor 1.0 0.0 0.0")
            assert green + red == 100 and 30 <= green <= 92
