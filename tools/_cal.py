import cocodr_amd, torch
from cocodr_amd import ops
torch.zeros(1).cuda()
print('event overhead us', [round(ops.prof_event_overhead_us(),2) for _ in range(5)])
