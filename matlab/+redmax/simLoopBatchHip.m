function [q, qdot, T, V, stats] = simLoopBatchHip(scene, itype, q0, qdot0, devices)
%simLoopBatchHip  simLoop of driverRedMaxBDF1.m:57-91 (itype 1) / driverRedMaxBDF2.m:57-125 (itype 2) for a BATCH of initial
% states of one scene, sharded over a list of MI355X - BASELINE.json's north_star ("the batch axis shards across GPUs") with
% MATLAB as the only host and no Python anywhere:
%
%   scene = scenesRedMax(sceneID);  scene.init();
%   [q, qdot, T, V, stats] = redmax.simLoopBatchHip(scene, 1, q0, qdot0, 0:7);     % q0, qdot0: nr x B
%
% q0(:,b), qdot0(:,b) is trajectory b (the DOF order of Joint.getQ).  The batch is split into numel(devices) contiguous
% shards; 'step' launches every shard's kernel before it waits for the first (rmx_group_step), so the devices run concurrently,
% and the per-step energies and the counters come back gathered into whole-batch arrays: T, V are B x nsteps, stats is B x 3
% [newton iterations, line-search halvings, RMX_ST_* bits].
if nargin < 5, devices = 0; end
B = size(q0, 2);
sim = redmax.HipSim(scene, B, devices);
guard = onCleanup(@() delete(sim));
sim.setState(q0, qdot0);
[T, V, stats] = sim.step(itype, scene.h, scene.nsteps);
[q, qdot] = sim.getState();
nbad = nnz(bitand(stats(:,3), 3));     % Newton diverged / did not converge (driverRedMaxBDF1.m:118-121, 150-153)
if nbad > 0
	fprintf('%d of %d trajectories had a Newton failure\n', nbad, B);
end
[wall, kernel] = sim.timing();
fprintf('%d trajectories x %d steps on %d shard(s): %.2f ms wall, slowest kernel %.2f ms\n', B, scene.nsteps, numel(kernel), wall, max(kernel));
end
