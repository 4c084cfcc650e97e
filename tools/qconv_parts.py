import sys, time, torch
sys.path.insert(0, '.')
from bench import Job
dev = torch.device('cuda', 0)
job = Job('qconv', torch.bfloat16, dev, None, 0)
def per_call(fn, n=3000):
    for _ in range(100): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6
for rep in range(2):
    for name, (x, g, q) in zip(['act [128,1024,14,14] per-tensor MAX', 'w [256,1024,1,1]', 'w [256,256,3,3]', 'w [1024,256,1,1]'], job.acts + job.weights):
        def step():
            x.grad = None
            q(x)[0].backward(g)
        def fwd():
            with torch.no_grad():
                q(x)
        print('%-40s step %.1f us   forward(no_grad) %.1f us' % (name, per_call(step), per_call(fwd)), flush=True)
    print('whole job.step %.1f us' % per_call(job.step, 1000))
