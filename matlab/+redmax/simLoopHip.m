function simLoopHip(scene, itype)
%simLoopHip  simLoop of driverRedMaxBDF1.m:57-91 (itype 1) / driverRedMaxBDF2.m:57-125 (itype 2) as ONE call into the
% HIP library: every step of the rollout runs on the device, then the per-step record that Scene.saveHistory keeps
% (Scene.m:134-161: q, qdot, T, V, t) is filled in and the final state goes back onto the joints.

jroot = scene.joints{1};
sim = redmax.HipSim(scene, 1);
guard = onCleanup(@() delete(sim));
[q0, qdot0] = jroot.getQ();
sim.setState(q0, qdot0);
[T, V, stats, Q, Qdot, C] = sim.step(itype, scene.h, scene.nsteps);   % C: Euler charts after every step (nsph x 1 x nsteps)

% Newton's messages (driverRedMaxBDF1.m:118-121, 150-153); bits: include/redmax_hip.h RMX_ST_*
if bitand(stats(1,3), 1), fprintf('Newton diverged\n'); end
if bitand(stats(1,3), 2), fprintf('Newton did not converge\n'); end

% JointSpherical / JointFree3D: q is expressed in the Euler chart the device ended in
if sim.nsph > 0
	charts = sim.getCharts();
	s = 0;
	for i = 1 : numel(scene.joints)
		if isprop(scene.joints{i}, 'chart')
			s = s + 1;
			scene.joints{i}.chart = double(charts(s,1));
		end
	end
end

replay = scene.drawHz > 0;
for k = 1 : scene.nsteps
	scene.t = k*scene.h;
	scene.k = k;
	scene.history(k).q = Q(:,1,k);
	scene.history(k).qdot = Qdot(:,1,k);
	if scene.computeH
		scene.history(k).T = T(1,k);
		scene.history(k).V = V(1,k);
		scene.history(k).t = scene.t;
	end
	if sim.nsph > 0
		scene.history(k).charts = double(C(:,1,k));   % the chart q and qdot of this step are expressed in
	end
	if replay
		s = 0;
		for i = 1 : numel(scene.joints)
			if isprop(scene.joints{i}, 'chart')
				s = s + 1;
				scene.joints{i}.chart = double(C(s,1,k));
			end
		end
		jroot.setQ(Q(:,1,k), Qdot(:,1,k));
		jroot.update();
		scene.draw();
	end
end
[q, qdot] = sim.getState();
jroot.setQ(q, qdot);
jroot.update();
scene.solverInfo = struct('newton_iters', double(stats(1,1)), 'ls_halvings', double(stats(1,2)), 'status', double(stats(1,3)));
end
