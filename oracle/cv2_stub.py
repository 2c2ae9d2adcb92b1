"""A stand-in ``cv2`` module built from oracle/cvleaves.py, so the GENUINE reference OA-Mix can be executed in
this container (tests/golden/refload.py installs it as sys.modules['cv2']).  Test infrastructure only."""
import types

from . import cvleaves as cv


class _SpectralResidual:
    def computeSaliency(self, image):
        return True, cv.spectral_residual_saliency(image)


def make_cv2(saliency_fn=None):
    m = types.ModuleType('cv2')
    m.warpAffine = cv.warp_affine
    m.getRotationMatrix2D = cv.get_rotation_matrix_2d
    m.GaussianBlur = lambda src, ksize, sigmaX=0, sigmaY=0, **k: cv.gaussian_blur(src, ksize, sigmaX, sigmaY)
    m.resize = lambda src, dsize, **k: cv.resize(src, dsize)
    sal = types.ModuleType('cv2.saliency')
    sal.StaticSaliencySpectralResidual_create = lambda: _SpectralResidual()
    m.saliency = sal
    m.INTER_LINEAR = 1
    m.BORDER_CONSTANT = 0
    return m
