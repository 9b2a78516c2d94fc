"""Permutation / plumbing layers with the reference's names and semantics
(/root/reference/nisrep/normalizing_flows/layers/layers.py).

Inside a `FlowSequential` these modules never execute: the engine folds them into the plane
indices the HIP kernels read and write (`remap`).  Called on their own (a user indexing the
Sequential) they are plain column shuffles done with torch indexing on whatever device the tensor
lives on -- host-side plumbing, no arithmetic.
"""
import numpy as np
import torch


class MaskLayer(torch.nn.Module):
    """layers.py:6-32: reorder columns to [feeder | trafoer | jacobian]; bit floor(pos/2) of the
    dimension index selects the pass-through ("feeder") set, pos%2 its value."""

    def __init__(self, dims_bin, pos, dev=None):
        super().__init__()
        feed = pos % 2
        masker = torch.as_tensor(dims_bin)[:, int(np.floor(pos / 2))]
        self.feeder = (masker == feed).nonzero()
        self.trafoer = (masker != feed).nonzero()
        if dev is not None:
            self.feeder = self.feeder.to(dev)
            self.trafoer = self.trafoer.to(dev)
        self.pass_through = self.feeder.shape[0]

    def order(self):
        return [int(i) for i in self.feeder.view(-1)] + [int(i) for i in self.trafoer.view(-1)]

    def remap(self, cur):
        return [cur[i] for i in self.order()]

    def forward(self, tensor):
        idx = torch.as_tensor(self.order() + [tensor.shape[1] - 1], device=tensor.device)
        return tensor.index_select(-1, idx)


class DeMaskLayer(torch.nn.Module):
    """layers.py:34-51: scatter the columns back to their original positions."""

    def __init__(self, first, second):
        super().__init__()
        self.list_ind = torch.unsqueeze(torch.squeeze(torch.cat((first, second), 0), -1), 0)

    def order(self):
        return [int(i) for i in self.list_ind.view(-1)]

    def remap(self, cur):
        new = list(cur)
        for j, target in enumerate(self.order()):
            new[target] = cur[j]
        return new

    def forward(self, tensor):
        ret = torch.empty_like(tensor)
        idx = torch.as_tensor(self.order(), device=tensor.device)
        ret[:, idx] = tensor[:, :-1]
        ret[:, -1] = tensor[:, -1]
        return ret


class Reshape(torch.nn.Module):
    """layers.py:55-64 (last module of a conditioner)."""

    def __init__(self, shapes1, shapes2):
        super().__init__()
        self.shapes = (shapes1, shapes2)

    def forward(self, tensor):
        return torch.reshape(tensor.clone(), (tensor.shape[0], self.shapes[0], self.shapes[1]))


class AddJacobian(torch.nn.Module):
    """layers.py:66-77: append the unit-Jacobian column (float64) -> (N, d+1)."""

    def __init__(self, jacobian_value=torch.ones(1)):
        super().__init__()
        self.jacobian_value = jacobian_value

    def forward(self, input, dev=torch.device("cpu")):
        return torch.cat((input.to(dev), self.jacobian_value.expand(input.shape[0], 1).to(dev).to(torch.double)), dim=1)


class RollLayer(torch.nn.Module):
    """layers.py:80-91: cyclic shift of the d variable columns, Jacobian column stays last."""

    def __init__(self, shift):
        super().__init__()
        self.shift = shift

    def remap(self, cur):
        d = len(cur)
        new = list(cur)
        for j in range(d):
            new[(j + self.shift) % d] = cur[j]
        return new

    def forward(self, x):
        return torch.cat((torch.roll(x[:, :-1], self.shift, dims=-1), x[:, -1:]), axis=-1)
