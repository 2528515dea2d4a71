"""Constants of the SVGTensor numeric format that the model and the loss depend on.

Restated from deepsvg/difflib/tensor.py:8-41 (command vocabulary :10, CMD_ARGS_MASK :15-21, argument column
indices :34-41).  Only the constants are mirrored: the tensor-building helpers of SVGTensor belong to the data
pipeline, which the reference keeps (deepsvg/svgtensor_dataset.py is used unchanged).
"""
import torch

#                       0    1    2    3     4      5     6
COMMANDS_SIMPLIFIED = ["m", "l", "c", "a", "EOS", "SOS", "z"]
M_ID, L_ID, C_ID, A_ID, EOS_ID, SOS_ID, Z_ID = range(7)

# which of the 11 argument slots (rx, ry, phi, fA, fS, qx1, qy1, qx2, qy2, x, y) each command uses
CMD_ARGS_MASK = torch.tensor([[0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1],   # m
                              [0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1],   # l
                              [0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1],   # c
                              [1, 1, 1, 1, 1, 0, 0, 0, 0, 1, 1],   # a
                              [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0],   # EOS
                              [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0],   # SOS
                              [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]])  # z

PAD_VAL = -1


class IndexArgs:
    RADIUS = slice(0, 2)
    X_AXIS_ROT = 2
    LARGE_ARC_FLG = 3
    SWEEP_FLG = 4
    CONTROL1 = slice(5, 7)
    CONTROL2 = slice(7, 9)
    END_POS = slice(9, 11)
