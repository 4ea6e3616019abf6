"""Calibration constants shipped with the reference (values of the 8 XML files that
7/triangulation.cpp:152-168,1069-1083 reads; millimetres, world frame = calibration board Z=0).
They are data, used as the realistic default rig for synthetic captures."""
import numpy as np

REF_CAM_SIZE = (1600, 1200)    # Camera_imagewidth x Camera_imageheight, PROJECT_GLOBAL/global_cv.h:49-50
REF_PROJ_SIZE = (1280, 720)    # Projector_imagewidth x Projector_imageheight, global_cv.h:52-53

REF_CALIBRATION = {
    "Kc": (1411.448307260438, 0.0, 793.9632451499745, 0.0, 1418.187495746432, 591.6074828524252, 0.0, 0.0, 1.0),
    "dc": (0.0813342523391422, -0.1102401608924523, 0.0, 0.0, 0.0),
    "rc": (-0.019872445941139634, -0.28645491955945396, 0.003766068255749373),
    "tc": (-50.78169517091686, -39.22827842204189, 103.28544711362919),
    "Kp": (2538.405763, 0.0, 580.56812, 0.0, 2487.799514191413, 485.0516260251576, 0.0, 0.0, 1.0),
    "dp": (0.0, 0.0, 0.0, 0.0, 0.0),
    "rp": (-0.0274370358554159, 0.4214701298993157, -0.016242416538549367),
    "tp": (-54.6742443377351, -38.36075907057487, 220.59730602483845),
}

CAL_KEYS = ("Kc", "dc", "rc", "tc", "Kp", "dp", "rp", "tp")


def scaled_calibration(W, H, PW, PH):
    """The reference rig rescaled to another camera / projector resolution: focal lengths and
    principal points scale with the image size, distortion and extrinsics stay."""
    cal = {k: np.array(v, dtype=np.float64) for k, v in REF_CALIBRATION.items()}
    sx, sy = W / REF_CAM_SIZE[0], H / REF_CAM_SIZE[1]
    px, py = PW / REF_PROJ_SIZE[0], PH / REF_PROJ_SIZE[1]
    Kc = cal["Kc"].reshape(3, 3).copy(); Kc[0] *= sx; Kc[1] *= sy
    Kp = cal["Kp"].reshape(3, 3).copy(); Kp[0] *= px; Kp[1] *= py
    cal["Kc"], cal["Kp"] = Kc.ravel(), Kp.ravel()
    return cal
