import sys
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo/efficient-slowfast_amd")
import microbench_conv as m
m.run("s2.a 64->64 @56", 8, 8, 56, 56, 64, 64)
m.run("s2.c 64->256 +res", 8, 8, 56, 56, 64, 256, res=True)
m.run("s2.a' 256->64", 8, 8, 56, 56, 256, 64)
m.run("s3.a 288->128 @56", 8, 8, 56, 56, 288, 128)
m.run("s3.c 128->512 +res", 8, 8, 28, 28, 128, 512, res=True)
m.run("s3.a' 512->128 @28", 8, 8, 28, 28, 512, 128)
m.run("s4.c 256->1024 +res", 8, 8, 14, 14, 256, 1024, res=True)
m.run("s4.a' 1024->256 1x1", 8, 8, 14, 14, 1024, 256)
m.run("s5.c 512->2048 +res", 8, 8, 7, 7, 512, 2048, res=True)
m.run("s5.a' 2048->512 1x1", 8, 8, 7, 7, 2048, 512)
