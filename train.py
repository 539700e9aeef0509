#!/usr/bin/env python3
"""Multi-GPU counterpart of `python3 airfoil_dqn.py` (reference airfoil_dqn.py:343-514):

    python train.py --config configs/ray_ys930.yaml                                   # 1 GPU
    python train.py --gpus 8 --config configs/ray_ys930.yaml                          # 8 GPUs (RCCL): starts its ranks itself
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py --config ...   # the same under torchrun

One process per GPU; every rank steps `--envs` environments (VecEnv2DAirfoil), all ranks apply the same
all-reduced gradient, replay transitions are optionally all-gathered.  The yaml is the reference's own format
(`flow_config`, `agent_params`, `optimizer`, `epsilon`); `geometry_params.mesh` may point at an .xdmf or .npz file."""
import argparse
import os as _os

# single-threaded BLAS / OpenMP pools BEFORE numpy / scipy / torch are imported: the hosts expose hundreds of logical CPUs
# under a small CPU quota, and idle pool threads that keep spinning starve the threads that launch kernels
for _k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    _os.environ.setdefault(_k, "1")
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # concurrent streams on separate hardware queues (meshdqn_amd/__init__.py)
import os

import numpy as np
import yaml


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--gpus", type=int, default=0,
                    help="N > 1 without a launcher (WORLD_SIZE unset): this process becomes the parent of N rank processes "
                         "(meshdqn_amd/launcher.py: 127.0.0.1 rendezvous, every rank watched, a dying rank ends the job)")
    ap.add_argument("--digest", action="store_true",
                    help="debugging / tests: one batched step per chunk, and after EVERY batched step every rank appends "
                         "sha256 digests of both Q-networks, the optimiser moments and its replay ring to "
                         "<save-dir>/digest_rank<r>.jsonl (replicas that diverge show up as differing lines)")
    ap.add_argument("--envs", type=int, default=128, help="environments per GPU")
    ap.add_argument("--steps", type=int, default=1000, help="batched rollout steps")
    ap.add_argument("--flow-steps", type=int, default=0, help="IPCS steps on the coarsened mesh per env step (S3)")
    ap.add_argument("--share-replay", action="store_true")
    ap.add_argument("--save-dir", default="training_results/run")
    ap.add_argument("--restart", action="store_true",
                    help="continue the run found in --save-dir: both Q-networks, optimiser / scheduler state, epsilon "
                         "counters and the logs (files of restart n carry n 'restart_' / 'RESTART_' prefixes, like the "
                         "reference's RESTART_NUM chain, airfoil_dqn.py:359-366)")
    ap.add_argument("--save-every", type=int, default=50, help="batched steps between checkpoints / log writes (0 = only at the end)")
    ap.add_argument("--host-loop", action="store_true",
                    help="host-driven loop (train_loop_vec: autograd replayed as a HIP graph, one read-back per step) instead "
                         "of the device-resident one (train_loop_device: replay, sampling, forward + backward and Adam as kernels)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import importlib.util
        import sys
        spec = importlib.util.spec_from_file_location("mdq_launcher", os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                                                    "meshdqn_amd", "launcher.py"))
        launcher = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(launcher)              # (by file path: the parent never imports the package / torch)
        have = launcher.visible_gpus()
        if have is not None and have < args.gpus and not os.environ.get("MDQ_SHARE_GPU"):
            raise SystemExit(f"--gpus {args.gpus} needs {args.gpus} GPUs on this node, the driver exposes {have} "
                             "(MDQ_SHARE_GPU=1 MDQ_DIST_BACKEND=gloo lets several ranks share a GPU, for debugging only)")
        rc, out0 = launcher.start_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus, tag="train")
        sys.stdout.write("".join(out0))
        raise SystemExit(rc)
    if args.gpus > 1 and int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} and the launcher's WORLD_SIZE={os.environ['WORLD_SIZE']} disagree")
    from meshdqn_amd.env import Env2DAirfoil
    from meshdqn_amd.trainer import DistContext, DQNTrainer, TrainingLog, train_loop_device, train_loop_vec
    from meshdqn_amd.vec_env import VecEnv2DAirfoil
    import torch
    torch.set_num_threads(1)   # (many-core hosts under a CPU quota: the CPU-side tensor ops are tiny, no intra-op pool)
    cfg = yaml.safe_load(open(args.config))
    ctx = DistContext()
    opt = cfg.get("optimizer", {})          # reference yaml sections: optimizer / epsilon (configs/ray_ys930.yaml)
    eps = cfg.get("epsilon", {})
    ap_ = cfg["agent_params"]
    trainer = DQNTrainer(n_actions=int(ap_["N_closest"]), num_inputs=2 + 3 * (int(ap_["solver_steps"]) // int(ap_["save_steps"])),
                         ctx=ctx, lr=float(opt.get("lr", 1e-5)), weight_decay=float(opt.get("weight_decay", 1e-6)),
                         batch_size=int(opt.get("batch_size", 32)), gamma=float(eps.get("gamma", 1.0)),
                         target_update=int(ap_.get("target_update", 50)))
    # restart chain: restart n reads the checkpoint with n - 1 "restart_" prefixes and writes with n (every rank loads the
    # same files, so the replicas stay identical)
    restart_num, steps_done0 = 0, None
    if args.restart:
        restart_num = sum(f.endswith("policy_net_1.pt") for f in os.listdir(args.save_dir))
        if restart_num == 0:
            raise SystemExit(f"--restart: no policy_net_1.pt checkpoint in {args.save_dir}")
        extra = trainer.load(args.save_dir, "restart_" * (restart_num - 1))
        steps_done0 = extra.get("steps_done")
        if steps_done0 is not None and len(steps_done0) != args.envs:      # other batch size: restart the counters at the mean
            steps_done0 = np.full(args.envs, int(np.mean(steps_done0)), np.int64)
    prefix = "restart_" * restart_num
    base = Env2DAirfoil(cfg, compute_device=ctx.device)          # ground truth + snapshots (the reference's first reset())
    venv = VecEnv2DAirfoil(cfg, args.envs, compute_device=ctx.device, base_env=base, flow_steps=args.flow_steps,
                           flow_overlap=args.flow_steps > 0)
    log = TrainingLog(args.save_dir, restart=args.restart, restart_num=restart_num) if ctx.rank == 0 else None

    def checkpoint(step, steps_done):
        if ctx.rank == 0:                                        # (the reference writes after every episode)
            trainer.save(args.save_dir, prefix, extra=dict(steps_done=np.asarray(steps_done).copy(), batched_steps=step))
            log.write()                                          # reward / rewards / losses / actions / eps .npy

    device_loop = ctx.device.type == "cuda" and venv.gpu_remesh and not args.host_loop
    loop = train_loop_device if device_loop else train_loop_vec
    kw, every, on_every = {}, args.save_every, checkpoint
    if args.digest:
        import hashlib
        import json
        os.makedirs(args.save_dir, exist_ok=True)
        dpath = os.path.join(args.save_dir, f"{prefix}digest_rank{ctx.rank}.jsonl")
        open(dpath, "w").close()

        def sha(tensors):
            h = hashlib.sha256()
            for t in tensors:
                h.update(t.detach().contiguous().cpu().numpy().tobytes())
            return h.hexdigest()

        def ring_rows(step):
            # FINISHED records of this job (the rows every rank must agree on): a step's records get their next state - and,
            # with the shared replay, travel to the other ranks - one step later (at once for the last step of the job)
            wrec = args.envs * (ctx.world if (args.share_replay and ctx.multi) else 1)
            return min((step if step >= args.steps else step - 1) * wrec, trainer.device_memory.capacity)

        def on_every(step, steps_done):          # noqa: F811 - every batched step: digests, then the periodic checkpoint
            torch.cuda.synchronize() if ctx.device.type == "cuda" else None
            rep = getattr(trainer, "device_memory", None)
            opt_state = [v for o in trainer.opts for st in o.state.values() for k, v in sorted(st.items()) if torch.is_tensor(v)]
            rec = dict(step=int(step), net1=sha(trainer.policy_net_1.parameters()), net2=sha(trainer.policy_net_2.parameters()),
                       optimiser=sha(opt_state), optimiser_steps=len(trainer.losses),
                       ring=None if rep is None else sha([rep.R[:ring_rows(step)]]), steps_done=[int(v) for v in steps_done])
            with open(dpath, "a") as f:
                f.write(json.dumps(rec) + "\n")
            if args.save_every and step % args.save_every == 0:
                checkpoint(step, steps_done)
        every = 1
        if device_loop:
            kw["chunk"] = 1
    out = loop(trainer, venv, args.steps, log=log, eps_decay=float(eps.get("decay", 10000)),
               eps_start=float(eps.get("start", 1.0)), eps_end=float(eps.get("end", 0.01)),
               share_replay=args.share_replay, steps_done0=steps_done0, every=every, on_every=on_every, **kw)
    if ctx.rank == 0:
        os.makedirs(args.save_dir, exist_ok=True)
        checkpoint(args.steps, out["steps_done"])
        np.save(os.path.join(args.save_dir, prefix + "step_rewards.npy"), out["rewards"])
        yaml.safe_dump(cfg, open(os.path.join(args.save_dir, "config.yaml"), "w"))
        print(f"ranks {ctx.world}: {args.steps} batched steps x {args.envs} envs/rank, mean reward {out['rewards'].mean():.4f}")
        if ctx.backend:
            print(f"process group: backend {ctx.backend}, {ctx.world} rank(s)")
    ctx.close()


if __name__ == "__main__":
    main()
