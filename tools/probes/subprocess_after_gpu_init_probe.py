import subprocess, sys, torch
torch.cuda.init(); x = torch.ones(4, device='cuda'); print('gpu up', float(x.sum()))
r = subprocess.run([sys.executable, '-c', 'print("child ok")'], capture_output=True, text=True, timeout=60)
print('child rc', r.returncode, r.stdout.strip(), r.stderr.strip()[-300:])
