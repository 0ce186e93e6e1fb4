"""Model zoo (reference: deeplearning4j-zoo)."""
from .zoo_model import ZooModel  # noqa: F401
from .lenet import LeNet  # noqa: F401
from .resnet50 import ResNet50  # noqa: F401
from .zoo import (AlexNet, SimpleCNN, VGG16, VGG19, Darknet19, TinyYOLO, YOLO2, SqueezeNet, UNet, Xception,  # noqa: F401
                  TextGenerationLSTM, InceptionResNetV1, FaceNetNN4Small2, NASNet)
