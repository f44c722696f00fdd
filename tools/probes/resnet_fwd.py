import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from latent2im_amd import synth
from latent2im_amd.regressor import ResNet50
net = ResNet50(synth.resnet50_state(seed=300), device='cuda')
x = torch.randn(8, 3, 1024, 1024, device='cuda')
for rep in range(2):
    y = net(x)
torch.cuda.synchronize()
